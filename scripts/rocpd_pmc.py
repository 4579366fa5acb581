"""Per-kernel PMC totals from rocprofv3 rocpd databases (one counter per pass, as MI355X_MICROARCH.md §HBM
prescribes).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide
coalesced streaming read (16 B/lane, global_load and buffer_load..lds alike) -> doubled here for the LDS-DMA kernels."""
import json, sqlite3, sys
fetch_db, write_db, out = sys.argv[1], sys.argv[2], sys.argv[3]

def per_kernel(db):
    c = sqlite3.connect(db)
    q = """select s.kernel_name, d.grid_size_x, count(*), sum(e.value), sum(d.end - d.start)
           from rocpd_pmc_event e join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name"""
    return {r[0]: (r[2], r[3], r[4]) for r in c.execute(q)}

f, w = per_kernel(fetch_db), per_kernel(write_db)
rows = {}
for k in f:
    n, fk, ns = f[k]
    wk = w.get(k, (0, 0.0, 0))[1]
    dma = "_GLOBAL__N_1" in k and "at6native" not in k      # this repo's kernels: all loads are 16 B/lane (wide, coalesced)
    fetch_bytes = fk * 1024 * (2 if dma else 1)
    rows[k] = {"launches": n, "fetch_bytes_per_launch": fetch_bytes / n, "write_bytes_per_launch": wk * 1024 / n,
               "hbm_bytes_per_launch": (fetch_bytes + wk * 1024) / n, "avg_us": ns / n / 1e3, "fetch_x2_correction": dma}
json.dump(rows, open(out, "w"), indent=1)
for k, r in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k[:60]:60s} n={r['launches']:5d} fetch/launch {r['fetch_bytes_per_launch']/1e6:9.1f} MB  write/launch {r['write_bytes_per_launch']/1e6:8.1f} MB  avg {r['avg_us']:8.1f} us  -> {r['hbm_bytes_per_launch']/r['avg_us']/1e3:7.1f} GB/s")
