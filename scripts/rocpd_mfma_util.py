"""Matrix-pipe utilisation of the dominant kernels from SQ counters (rocprofv3 rocpd databases, one or more --pmc passes).

    python scripts/rocpd_mfma_util.py out.json pass1_results.db [pass2_results.db ...]

rocprofv3 reports one row per XCD (8 per dispatch on MI355X) for every counter; the per-launch figure of an SQ counter is the SUM over the
rows, GRBM_GUI_ACTIVE (chip cycles while the launch is resident) is the same on every row.  Units (MI355X_MICROARCH.md, "Per-instruction cycle
constants"; checked here against instruction counts): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES sums, over the 1024 SIMDs, the cycles a SIMD's matrix pipe is busy (16 per v_mfma_f32_16x16x32_bf16, 32 per
v_mfma_f32_32x32x16_bf16; = SQ_INSTS_VALU_MFMA_MOPS_BF16 / 2: for gemm_bf16_t256<0> the counter / 1024 / GRBM_GUI_ACTIVE reproduces
FLOPs / time / (2.5 PFLOP/s x clock / 2.4 GHz) to within 2 %, for the attention kernel to 0.5 %).  Derived:

    mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)   matrix pipe busy over the whole launch, at the clock it ran at
    clock_ghz   = GRBM_GUI_ACTIVE / launch duration (of the same profiled pass)
    sq_*_per_wave_cycle = counter / SQ_WAVE_CYCLES                             share of a resident wave's time: SQ_WAIT_ANY = parked at s_waitcnt /
                          s_barrier, SQ_WAIT_INST_ANY = stalled at issue (matrix pipe busy, dependencies), SQ_ACTIVE_INST_ANY = issuing

One JSON row per kernel; bench.py reads the newest profiles/rNN_pmc_mfma_util.json."""
import json, sqlite3, sys

out = sys.argv[1]
SIMDS = 256 * 4
rows = {}
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    def tab(prefix): return [t for t in tabs if t.startswith(prefix)][0]
    pmc, disp, sym, info = tab("rocpd_pmc_event"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_info_pmc")
    q = f"""select s.kernel_name, i.name, count(*), sum(e.value), count(distinct d.id), sum(d.end - d.start)
            from {pmc} e join {disp} d on e.event_id = d.event_id join {sym} s on d.kernel_id = s.id join {info} i on e.pmc_id = i.id
            group by s.kernel_name, i.name"""
    tag = db.rsplit("/", 1)[-1].split("_")[0]
    for k, n, cnt, val, nd, ns in c.execute(q):
        r = rows.setdefault(k, {"counters": {}, "launches": nd, "avg_us": {}, "instances": cnt // max(nd, 1)})
        key = n if n not in r["counters"] else f"{n}#{tag}"
        inst = cnt / max(nd, 1)
        r["counters"][key] = val / cnt if n.startswith("GRBM") else val / nd          # GRBM: per row; SQ: summed over the rows of a dispatch
        r["avg_us"][key] = ns / cnt / 1e3

result = {}
want = ("gemm_bf16_w4p", "gemm_bf16_t256", "attention_d128", "attention_q64", "gemm_bf16_t128", "gemm_bf16_deep", "qk_norm_rope", "layernorm_modulate", "conv2d_f32")
for k, r in sorted(rows.items(), key=lambda kv: -max(kv[1]["avg_us"].values()) * kv[1]["launches"]):
    if not any(w in k for w in want):
        continue
    c = r["counters"]
    d = {"launches": r["launches"], "rows_per_dispatch": r["instances"], "avg_us_profiled": r["avg_us"].get("GRBM_GUI_ACTIVE", max(r["avg_us"].values())),
         "counters_per_launch": c}
    gui = c.get("GRBM_GUI_ACTIVE")
    if gui:
        d["clock_ghz"] = gui / r["avg_us"]["GRBM_GUI_ACTIVE"] / 1e3
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            d["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS)
    wc = c.get("SQ_WAVE_CYCLES")
    wc2 = c.get("SQ_WAVE_CYCLES#pass2", wc)
    if wc:
        for n in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY",
                  "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM", "SQ_LDS_IDX_ACTIVE"):
            if n in c:
                base = wc if n in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT") else wc2      # (the pass the counter came from)
                d[n.lower() + "_per_wave_cycle"] = c[n] / base
    result[k] = d
    print(f"{k[:100]}\n   launches={d['launches']} ({d['rows_per_dispatch']} counter rows each) avg={d['avg_us_profiled']:.1f} us (profiled)"
          + (f" clock={d['clock_ghz']:.3f} GHz" if "clock_ghz" in d else "")
          + (f"  MFMA busy = {100 * d['mfma_util']:.1f} % of the launch's SIMD-cycles" if "mfma_util" in d else ""))
    for n, v in sorted(c.items()):
        extra = ""
        if wc and n.startswith("SQ_") and not n.startswith("SQ_WAVE_CYCLES"):
            extra = f"   = {v / wc:.4f} of SQ_WAVE_CYCLES"
        print(f"      {n:40s} {v:18.0f} /launch{extra}")
json.dump(result, open(out, "w"), indent=1)
