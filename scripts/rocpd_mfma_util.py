"""Matrix-pipe utilisation of the dominant kernels from SQ counters (rocprofv3 rocpd databases, one or more --pmc passes).

    python scripts/rocpd_mfma_util.py out.json pass1_results.db [pass2_results.db ...]

Units (MI355X_MICROARCH.md, "Per-instruction cycle constants"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_BUSY_CYCLES counts cycles summed over the SQs that
report (per shader engine); GRBM_GUI_ACTIVE counts chip cycles.  The derived figures:

    mfma_util         = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 256 CUs x 4 SIMDs)     matrix pipe busy, over the whole launch
    clock_ghz         = GRBM_GUI_ACTIVE / launch duration
    *_per_wave_cycle  = counter / SQ_WAVE_CYCLES                                              share of a resident wave's time

One JSON row per kernel; the bench line reads the newest profiles/rNN_pmc_mfma_util.json."""
import json, sqlite3, sys

out = sys.argv[1]
SIMDS = 256 * 4
rows = {}
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    def tab(prefix): return [t for t in tabs if t.startswith(prefix)][0]
    pmc, disp, sym, info = tab("rocpd_pmc_event"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_info_pmc")
    q = f"""select s.kernel_name, i.name, count(*), sum(e.value), sum(d.end - d.start)
            from {pmc} e join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id join {info} i on e.pmc_id = i.id
            group by s.kernel_name, i.name"""
    for k, n, cnt, val, ns in c.execute(q):
        r = rows.setdefault(k, {"counters": {}, "launches": cnt, "avg_us": {}})
        # a counter present in several passes (GRBM_GUI_ACTIVE): keep each pass's own duration beside it
        key = n if n not in r["counters"] else f"{n}#{db.rsplit('/', 1)[-1].split('_')[0]}"
        r["counters"][key] = val / cnt
        r["avg_us"][key] = ns / cnt / 1e3

result = {}
want = [a for a in ("gemm_bf16_t256", "attention_d128", "gemm_bf16_t128", "gemm_bf16_deep", "qk_norm_rope", "layernorm_modulate", "conv2d_f32")]
for k, r in sorted(rows.items(), key=lambda kv: -max(kv[1]["avg_us"].values()) * kv[1]["launches"]):
    if not any(w in k for w in want):
        continue
    c = r["counters"]
    d = {"launches": r["launches"], "avg_us_profiled": max(r["avg_us"].values()), "counters_per_launch": c}
    gui = c.get("GRBM_GUI_ACTIVE")
    if gui:
        d["clock_ghz"] = gui / r["avg_us"]["GRBM_GUI_ACTIVE"] / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS)
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for n in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT"):
            if n in c:
                d[n.lower() + "_per_wave_cycle"] = c[n] / wc
    result[k] = d
    print(f"{k[:90]}\n   launches={d['launches']} avg={d['avg_us_profiled']:.1f} us (profiled)"
          + (f" clock={d['clock_ghz']:.3f} GHz" if "clock_ghz" in d else "")
          + (f"  MFMA busy = {100 * d['mfma_util']:.1f} % of SIMD-cycles" if "mfma_util" in d else ""))
    for n, v in sorted(c.items()):
        print(f"      {n:40s} {v:18.0f} /launch" + (f"   = {v / wc:.4f} of SQ_WAVE_CYCLES" if wc and n.startswith("SQ_") and n != "SQ_WAVE_CYCLES" else ""))
json.dump(result, open(out, "w"), indent=1)
