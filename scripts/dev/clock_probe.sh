#!/bin/bash
# what a non-root user on the GPU box can read about clocks / power
echo "== env"; env | grep -i -E "ROCR|HIP_VIS|CUDA_VIS|GPU" | head
echo "== torch"; python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(p.name, getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None), getattr(p, "pci_domain_id", None), p.multi_processor_count, getattr(p, "clock_rate", None))
try:
    import amdsmi
    print("amdsmi importable", amdsmi.__file__)
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    print("handles", len(hs))
    for h in hs[:8]:
        try:
            print(amdsmi.amdsmi_get_gpu_device_bdf(h), amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX), amdsmi.amdsmi_get_power_info(h))
        except Exception as e:
            print("handle err", e)
except Exception as e:
    print("amdsmi:", type(e).__name__, e)
PY
echo "== rocm-smi"; timeout 20 rocm-smi --showclocks --showpower --json 2>&1 | head -c 1500
echo; echo "== sysfs"; ls /sys/class/drm/ 2>&1 | head -20
for c in /sys/class/drm/card*/device; do echo $c; cat $c/pp_dpm_sclk 2>&1 | head -4; ls $c/hwmon 2>/dev/null; for h in $c/hwmon/hwmon*; do cat $h/power1_average 2>&1 | head -1; cat $h/freq1_input 2>&1 | head -1; done; done 2>&1 | head -60
