import subprocess, torch, os
torch.cuda.init(); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
for cmd in (["/opt/rocm/bin/hipcc", "--version"], ["/bin/true"], ["python3", "-c", "print(1)"]):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        print(cmd[0], "rc", r.returncode, (r.stdout or r.stderr).strip().split("\n")[0][:80])
    except Exception as e:
        print(cmd[0], "EXC", type(e).__name__, e)
