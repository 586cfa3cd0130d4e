export TMPDIR=/tmp
for spb in 128 256 512; do
  rm -rf gpurun_out/spb_pmc_$spb
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/spb_pmc_$spb -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --spb $spb > /dev/null 2> gpurun_out/spb_pmc_$spb.err
  python3 - $spb <<'PY'
import csv, glob, sys
v = sys.argv[1]
for f in glob.glob(f'gpurun_out/spb_pmc_{v}/*counter_collection.csv'):
    rows = [r for r in csv.DictReader(open(f)) if 'forward_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
    fs = sum(float(r['Counter_Value']) for r in rows) / len(rows)
    dur = sum(float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in rows) / len(rows)
    print(f"spb {v}: kernel {dur/1e6:.2f} ms  FETCH_SIZE {fs:.4e} KiB -> HBM read {2*fs*1024/1e9:.1f} GB per launch")
PY
done
