"""profiles/pmc_traffic.json from the PMC passes of profiles/tools/gpu_profile_round.sh (summarize_pmc.py tables).
Per dominant kernel: HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; gfx950's FETCH_SIZE tallies 128-B
read requests at 64 B: MI355X_MICROARCH.md, HBM section) and, where the instruction-mix passes saw the kernel, VALU /
transcendental instructions per MFMA -- stamped with bench.kernel_src_sha() of the kernel sources it was measured on:
bench.py prints `traffic: null` for any other source state.
usage: python profiles/make_pmc_traffic.py <pmc_counters.md> [<pmc_instruction_mix.md>]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_src_sha  # noqa: E402

rows = {}
for path in sys.argv[1:]:
    for line in open(path):
        m = re.match(r"\| `(.+?)` \| ([A-Z_0-9]+) \| (\d+) \| ([0-9.e+]+) \|", line)
        if m:
            rows.setdefault(m.group(1), {})[m.group(2)] = float(m.group(4))
names = {"k_reni_train_bf16<128, true, false, false, true, true>": "k_reni_train_bf16<128,true,L0X>",   # (SPEC + L0X: what config 2 runs, round 5)
         "k_reni_l0_ring<128>": "k_reni_l0_ring",                                                         # (layer 0's backward + dW_1 behind it)
         "k_reni_train_bf16<128, true, false, false, true, false>": "k_reni_train_bf16<128,true>",       # (the SPEC instance of round 4: RENI_NO_L0X)
         "k_reni_train_bf16<128, false, false, false, false, false>": "k_reni_train_bf16<128,false>",
         "k_reni_train_bf16<128, false, true, false, false, false>": "k_reni_train_bf16<128,false,true>",
         "k_reni_train_bf16<128, true, false, true, true, false>": "k_reni_train_bf16<128,true,false,true>",   # (FiLM: its SPEC instance)
         "k_reni_main<reni::PolBF16, 256, 2, false>": "k_reni_main<bf16,H=256,FWD_BWD>",                 # (c2_h256: the shipped width)
         "k_reni_wide256<2, false>": "k_reni_wide256<2>",                                                  # (c2_h256 from round 5: the training form)
         "k_reni_wide256<1, false>": "k_reni_wide256<1>",                                                  # (c4 at H = 256: frozen decoder)
         "k_reni_wide256<0, false>": "k_reni_wide256<0>",                                                  # (forward / statistics at H = 256; round 6: <MODE, FILM>)
         "k_dw_frag<256, false>": "k_dw_frag<256>",
         "k_reni_main<reni::PolF32, 128, 0, false>": "k_reni_main<f32,H=128,FWD>"}
sha = kernel_src_sha()
out = {}
for k, v in rows.items():
    for frag, nice in names.items():
        if frag in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            rec = {"hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), "FETCH_SIZE_KB": v["FETCH_SIZE"],
                   "WRITE_SIZE_KB": v["WRITE_SIZE"], "src_sha256": sha,
                   "note": "2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes of bench.py --steps 3; L2 <-> fabric bytes, Infinity-Cache hits included"}
            if "SQ_INSTS_MFMA" in v:  # per launch: bench.py's frac_issued = this x FLOP per MFMA / the kernel's own time
                rec["mfma_per_launch"] = int(v["SQ_INSTS_MFMA"])
            if "SQ_INSTS_VALU" in v and "SQ_INSTS_MFMA" in v:  # (SQ_INSTS_VALU counts the MFMAs too)
                rec["valu_per_mfma"] = round((v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"], 3)
                if "SQ_INSTS_VALU_TRANS" in v:
                    rec["trans_per_mfma"] = round(v["SQ_INSTS_VALU_TRANS"] / v["SQ_INSTS_MFMA"], 3)
            out[nice] = rec
print(json.dumps(out, indent=1))
