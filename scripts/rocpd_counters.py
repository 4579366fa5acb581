"""Per-kernel totals of every PMC counter in a rocprofv3 rocpd database (one --pmc pass): counter sums per launch and
the kernel's average duration, plus derived effective clock when GRBM_GUI_ACTIVE is present."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
def tab(prefix): return [t for t in tabs if t.startswith(prefix)][0]
pmc, disp, sym, info = tab("rocpd_pmc_event"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_info_pmc")
q = f"""select s.kernel_name, i.name, count(*), sum(e.value), sum(d.end - d.start)
        from {pmc} e join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id join {info} i on e.pmc_id = i.id
        group by s.kernel_name, i.name"""
rows = {}
for k, n, cnt, val, ns in c.execute(q):
    rows.setdefault(k, {})[n] = (cnt, val, ns)
for k, d in rows.items():
    if not any(x in k for x in sys.argv[2:] or [""]): continue
    cnt, _, ns = next(iter(d.values()))
    us = ns / cnt / 1e3
    print(f"{k[:70]}  launches={cnt} avg={us:.1f} us")
    for n, (cc, val, _) in sorted(d.items()):
        per = val / cc
        extra = f"  -> {per / us / 1e3:.3f} GHz effective" if n == "GRBM_GUI_ACTIVE" else ""
        print(f"    {n:34s} {per:16.0f} /launch{extra}")
