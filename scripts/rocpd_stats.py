"""Summarise a rocprofv3 rocpd sqlite database: per-kernel calls / total / avg / share (like --stats).

    python scripts/rocpd_stats.py results.db [marker n]

With `marker n`: only the dispatches that START after the END of the n-th dispatch whose kernel name contains `marker` — e.g.
`image_postprocess 1` = everything after the first (warm-up) batch of bench.py, i.e. the timed region without model construction
(weight stacking and random initialisation launch hundreds of ATen cat / copy / fill kernels that never run per batch)."""
import sqlite3, sys
db = sys.argv[1]
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(rocpd_kernel_dispatch)")]
t_from = 0
if len(sys.argv) > 3:
    marker, n = sys.argv[2], int(sys.argv[3])
    ends = [r[0] for r in c.execute("""select d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                                       where s.kernel_name like ? order by d.start""", (f"%{marker}%",))]
    if len(ends) < n:
        sys.exit(f"only {len(ends)} dispatches of *{marker}* in {db}")
    t_from = ends[n - 1]
    print(f"# dispatches that start after the end of dispatch {n} of *{marker}* ({len(ends)} in the trace)")
rows = c.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end-d.start), max(d.end-d.start)
                    from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                    where d.start > ? group by s.kernel_name order by 3 desc""", (t_from,)).fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
for n, cnt, t, avg, mn, mx in rows[:(200 if t_from else 40)]:
    n = n if len(n) <= 70 else n[:67] + "..."
    print(f"{n:70s} {cnt:7d} {t/1e6:10.3f} {avg/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*t/tot:6.2f}")
print(f"TOTAL kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
