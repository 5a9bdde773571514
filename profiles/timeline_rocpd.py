"""One step's kernel timeline from a rocprofv3 rocpd database: start offset, duration, queue of every kernel between two
consecutive launches of the dominant kernel.  usage: python profiles/timeline_rocpd.py <results.db> [kernel substring]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "k_reni_train_bf16"
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute(f"select name, start, end{', ' + qcol if qcol else ''} from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if key in r[0]]
a, b = idx[-3], idx[-2]
t0 = rows[a][1]
print(f"step = {(rows[b][1] - t0) / 1e3:.1f} us between two launches of {key}")
for r in rows[a:b + 1]:
    print(f"{(r[1] - t0) / 1e3:9.1f} us  +{(r[2] - r[1]) / 1e3:8.1f} us  q{r[3] if qcol else '?'}  {r[0][:70]}")
