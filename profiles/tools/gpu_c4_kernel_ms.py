"""One line: config 4's step (dense: every tile + the statistics pass) -- step ms, frozen-kernel ms, statistics-pass ms.  For gpu_variants.sh --cmd."""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c4", "--dense", "--no-cpu-baseline", "--steps", "40", "--warmup", "20"] + sys.argv[1:],
                   capture_output=True, text=True)
try:
    d = json.loads(r.stdout.strip().splitlines()[-1])
    rf = d["roofline"]
    print("c4 dense: step ms %.4f  frozen kernel ms %.4f (min %.4f)  stats pass ms %.4f  frac %.4f" %
          (d["ms_per_step"], rf["kernel_avg_ms"], rf["kernel_min_ms"], rf.get("stats_pass_avg_ms", 0.0), rf["frac"]))
except Exception as e:  # noqa: BLE001
    print("FAILED", e, r.stderr[-500:])
