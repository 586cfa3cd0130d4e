export TMPDIR=/tmp
for v in base 1 4 8 15; do
  if [ $v = base ]; then unset BNN_CHAOS_SO; else export BNN_CHAOS_SO=$PWD/bnn_chaos_model_amd/csrc/libabl_$v.so; fi
  rm -rf gpurun_out/abl_pmc_$v
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/abl_pmc_$v -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/abl_pmc_$v.err
  python3 - $v <<'PY'
import csv, glob, sys
v = sys.argv[1]
for f in glob.glob(f'gpurun_out/abl_pmc_{v}/*counter_collection.csv'):
    rows = [r for r in csv.DictReader(open(f)) if 'forward_kernel' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE']
    cyc = sum(float(r['Counter_Value']) for r in rows) / len(rows) / 8
    dur = sum(float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in rows) / len(rows)
    print(f"ablate {v}: kernel {dur/1e6:.2f} ms  cycles {cyc:.4e}  clock {cyc/dur:.4f} GHz  MFMA-only floor 1.1245e9 cycles -> busy {1.1245e9/cyc:.4f}")
PY
done
