import json, subprocess, sys
res = {}
for k in (10, 20, 40, 80, 160):
    out = subprocess.run([sys.executable, "bench.py", "--steps", str(k), "--warmup", "5", "--windows", "9", "--no-cpu-baseline",
                          "--no-probe", "--no-moments", "--no-validate"], capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    res[k] = d["ms_per_step"] * k
    print(k, round(d["ms_per_step"], 4), [round(x, 4) for x in d["ms_per_step_windows"]], flush=True)
ks = sorted(res)
import numpy as np
A = np.vstack([np.ones(len(ks)), ks]).T
a, b = np.linalg.lstsq(A, np.array([res[k] for k in ks]), rcond=None)[0]
print("fit: window_ms = %.4f + %.4f * steps" % (a, b))
