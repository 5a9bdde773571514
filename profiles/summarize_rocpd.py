"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel stats table (markdown).
usage: python profiles/summarize_rocpd.py <results.db> [out.md [K [W]]]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, cnt, s, a, mn, mx in rows:
        out.append(f"| `{n[:110]}` | {cnt} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.1f} |")
    # the default bench command: the headline's W warm-up + K timed steps come FIRST (arguments 3, 4 = K, W), the sub-records behind
    # them, the same W + K steps once more at the end (`sustained`).  The average of launches W .. W+K-1 of the training kernel is what
    # the bench line's roofline.kernel_avg_ms must agree with; the last K launches are the sustained window.
    if len(sys.argv) > 3:
        k = int(sys.argv[3])
        w = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        # (prefixes: the training instance has a sixth template argument since round 5 -- L0X -- and the exact round-4 name matched nothing)
        for key in ("k_reni_train_bf16<128, true, false, false, true", "k_reni_l0_ring<128>", "k_reni_dw1_ring<128"):
            allr = [r[0] for r in c.execute(f"select end-start from kernels where {name_col} like '%{key}%' order by start").fetchall()]
            for title, d in (("headline window (launches %d..%d)" % (w, w + k - 1), allr[w:w + k]), ("sustained window (the last %d launches)" % k, allr[-k:])):
                if d:
                    out.append("")
                    out.append(f"{title} of `{key}...`: avg {sum(d)/len(d)/1e3:.1f} us, min {min(d)/1e3:.1f} us, max {max(d)/1e3:.1f} us")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
