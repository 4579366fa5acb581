"""Per-variant fabric traffic and effective clock of the 256x256 GEMM from ONE rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE pass over
scripts/pmc_gemm_walk.py:   python scripts/rocpd_gemm_walk.py results.db plan.json out.json

The dispatches of gemm_bf16_t256 are taken in start order and cut into the plan's chunks; the first launch of a chunk is dropped (it
finds the previous variant's panels in the caches).  FETCH_SIZE is in KiB and, on gfx950, reports half the bytes of 16-byte-per-lane
streaming reads (MI355X_MICROARCH.md, HBM): doubled here — every load of this kernel is such a read.  It counts the L2s' fabric-side
read requests, Infinity-Cache hits included: "bytes the eight L2s pulled", not HBM bytes.  clock = GRBM_GUI_ACTIVE / duration."""
import json, sqlite3, sys
db, plan_path, out = sys.argv[1], sys.argv[2], sys.argv[3]
plan = json.load(open(plan_path))
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
def tab(prefix): return [t for t in tabs if t.startswith(prefix)][0]
pmc, disp, sym, info = tab("rocpd_pmc_event"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_info_pmc")
q = f"""select d.id, d.start, d.end - d.start, i.name, sum(e.value), count(*)
        from {pmc} e join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id join {info} i on e.pmc_id = i.id
        where (s.kernel_name like '%gemm_bf16_t256%' or s.kernel_name like '%gemm_bf16_w4p%') group by d.id, i.name order by d.start"""
launches = {}
for did, start, ns, name, val, cnt in c.execute(q):
    r = launches.setdefault(did, {"start": start, "ns": ns})
    r[name] = val / cnt if name.startswith("GRBM") else val
rows = sorted(launches.values(), key=lambda r: r["start"])
need = sum(p["launches"] for p in plan)
assert len(rows) == need, f"{len(rows)} dispatches of gemm_bf16_t256 / gemm_bf16_w4p in the database, the plan has {need}"
res, k = [], 0
for p in plan:
    chunk = rows[k + 1:k + p["launches"]]; k += p["launches"]
    n = len(chunk)
    us = sum(r["ns"] for r in chunk) / n / 1e3
    fetch = 2 * 1024 * sum(r["FETCH_SIZE"] for r in chunk) / n
    clk = sum(r["GRBM_GUI_ACTIVE"] / r["ns"] for r in chunk) / n
    M, N, K = p["M"], p["N"], p["K"]
    algo = 2 * (M * K + N * K)
    res.append(dict(p, us=us, tflops=2 * M * N * K / us / 1e6, clock_ghz=clk, fetch_bytes=fetch, fetch_over_operands=fetch / algo))
json.dump(res, open(out, "w"), indent=1)
last = None
for r in res:
    if (r["M"], r["N"], r["K"]) != last:
        last = (r["M"], r["N"], r["K"])
        tm, tn = (r["M"] + 255) // 256, (r["N"] + 255) // 256
        print(f"M={r['M']} N={r['N']} K={r['K']}  ({tm} x {tn} tiles; operands {2 * (r['M'] + r['N']) * r['K'] / 1e6:.0f} MB, output {2 * r['M'] * r['N'] / 1e6:.0f} MB)")
    print(f"   mfma {r['mfma']} nt {r.get('store_nt', 0)} group_m {r['group_m']:3d}: {r['us']:8.1f} us  {r['tflops']:7.0f} TFLOP/s  clock {r['clock_ghz']:.3f} GHz  "
          f"fetched {r['fetch_bytes'] / 1e9:6.2f} GB = {r['fetch_over_operands']:5.2f} x the operands")
