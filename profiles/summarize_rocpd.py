"""Summarise a rocprofv3 rocpd database (kernel trace) into a per-kernel stats table (markdown).
usage: python profiles/summarize_rocpd.py <results.db> [out.md [K]]"""
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
    # the default bench command runs its sub-records first: the headline's K timed steps are the LAST K launches of the training
    # kernel in the process (argument 3 = K) -- their average is what the bench line's kernel_avg_ms must agree with
    if len(sys.argv) > 3:
        k = int(sys.argv[3])
        last = c.execute(f"select end-start from kernels where {name_col} like '%k_reni_train_bf16<128, true, false, false, true>%' "
                         f"order by start desc limit {k}").fetchall()
        if last:
            d = [r[0] for r in last]
            out.append("")
            out.append(f"headline window: the last {len(d)} launches of `k_reni_train_bf16<128, true, false, false, true>`: "
                       f"avg {sum(d)/len(d)/1e3:.1f} us, min {min(d)/1e3:.1f} us, max {max(d)/1e3:.1f} us")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
