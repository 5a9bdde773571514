"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of profiles/tools/gpu_profile_round.sh (summarize_pmc.py tables).
Per dominant kernel: HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; gfx950's FETCH_SIZE tallies 128-B
read requests at 64 B: MI355X_MICROARCH.md, HBM section), stamped with the sha256 of the kernel sources it was measured on --
bench.py prints `traffic: null` for any other source state.
usage: python profiles/make_pmc_traffic.py <pmc_counters.md>"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = {}
for line in open(sys.argv[1]):
    m = re.match(r"\| `(.+?)` \| (FETCH_SIZE|WRITE_SIZE) \| (\d+) \| ([0-9.e+]+) \|", line)
    if m:
        rows.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
names = {"k_reni_train_bf16<128, true, false, false>": "k_reni_train_bf16<128,true>",
         "k_reni_train_bf16<128, false, false, false>": "k_reni_train_bf16<128,false>",
         "k_reni_train_bf16<128, true, false, true>": "k_reni_train_bf16<128,true,false,true>"}
sha = hashlib.sha256(open(os.path.join(ROOT, "reni_amd", "csrc", "reni_device.inc"), "rb").read()).hexdigest()
out = {}
for k, v in rows.items():
    for frag, nice in names.items():
        if frag in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[nice] = {"hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), "FETCH_SIZE_KB": v["FETCH_SIZE"],
                         "WRITE_SIZE_KB": v["WRITE_SIZE"], "src_sha256": sha,
                         "note": "2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes of bench.py --steps 3; L2 <-> fabric bytes, Infinity-Cache hits included"}
print(json.dumps(out, indent=1))
