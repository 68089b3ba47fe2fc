import sys, os, time, glob, threading, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN
print([ (d, os.listdir(d)) for d in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")][:2])
ps = BN.PowerSampler(0.1)
print(ps.power, ps.freq)
x = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
with ps:
    t0 = time.time()
    while time.time() - t0 < 6:
        for _ in range(20):
            y = x @ x
        torch.cuda.synchronize()
print([(round(a - ps.samples[0][0], 1), round(p), round(f)) for a, p, f in ps.samples][::3])
os.system("rocm-smi --showpower --showclocks | grep -i 'sclk\\|Power'")
