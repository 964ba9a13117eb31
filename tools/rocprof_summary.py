#!/usr/bin/env python3
"""Summarise rocprofv3 output databases (rocpd sqlite, the default of ROCm 7.x) into the small CSVs kept under
profiles/.  Usage:
    tools/rocprof_summary.py stats  <trace_results.db>                  > profiles/rNN_kernel_stats.csv
    tools/rocprof_summary.py gaps   <trace_results.db>                  (GPU busy / idle per train step)
    tools/rocprof_summary.py timeline <trace_results.db> [step]         (one step: start offset, duration, gap before)
    tools/rocprof_summary.py hbm-csv <fetch_pass.csv> <write_pass.csv> <batch>   > profiles/rNN_pmc_hbm.csv
    tools/rocprof_summary.py pmc-csv <x_counter_collection.csv> [...]   > profiles/rNN_pmc.csv  (per-dispatch averages;
                                                                        from rocprofv3 --pmc ... --output-format csv)
    tools/rocprof_summary.py sq-derived <rNN_pmc_sq.csv>                (fractions of wave cycles, matrix-pipe busy share)
Counter passes are collected separately (rocprofv3 --kernel-trace --pmc A B ...; gpurun refuses --pmc with --stats)."""
import collections
import sqlite3
import sys


import re

# template instances that differ only in their workgroup width (waves per workgroup: 8, or 4 in the narrow mapping of small
# batches) share the name the per-kernel timing table of libdvs_hip.so and bench.py use
_WIDTH_ONLY = re.compile(r"^(k_ffn_bwd|k_attn_bwd|k_attn_fwd)<\d+>$")
_KEEP_FIRST = re.compile(r"^(k_bwd_stack|k_fwd_stack|k_proj_bwd)<(\d+), *\d+>$")


def short(name):
    name = name.split("(")[0].replace("void ", "")
    m = _KEEP_FIRST.match(name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = _WIDTH_ONLY.match(name)
    if m:
        return m.group(1)
    if name.startswith("_Z"):            # a name rocprofv3 could not demangle (_Float16-style types in the signature): _Z<len><name>...
        i = 2
        while i < len(name) and name[i].isdigit():
            i += 1
        if i > 2:
            name = name[i:i + int(name[2:i])]
    return name


def kernels(db):
    cur = sqlite3.connect(db).cursor()
    return list(cur.execute("select name, start, end from kernels order by start"))


def stats(db, skip_frac=0.0):
    """Per-kernel statistics; skip_frac drops the first share of every kernel's launches (the warm-up steps of a bench run: the
    clocks and caches of a fresh process settle over the first ~15 steps, bench.py's timed region starts behind its own warm-up)."""
    agg = collections.defaultdict(list)
    for name, s, e in kernels(db):
        agg[short(name)].append(e - s)
    if skip_frac > 0.0:
        agg = {k: v[int(len(v) * skip_frac) & ~1:] or v for k, v in agg.items()}      # (an even number: chained kernels launch in pairs)
    total = sum(sum(v) for v in agg.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f'"{k}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)}')


def timeline(db, step=8):
    rows = kernels(db)
    names = [short(r[0]) for r in rows]
    idx = [i for i, n in enumerate(names) if n in ("k_pack", "k_pack_w")]
    a, b = idx[step], idx[step + 1]
    t0 = rows[a][1]
    prev_end = rows[a - 1][2] if a > 0 else t0
    print("kernel,start_us,duration_us,gap_before_us")
    for name, s, e in rows[a:b + 1]:
        print(f"{short(name)[:60]},{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{(s - prev_end) / 1e3:.1f}")
        prev_end = max(prev_end, e)


def gaps(db):
    rows = kernels(db)
    names = [short(r[0]) for r in rows]
    idx = [i for i, n in enumerate(names) if n in ("k_pack", "k_pack_w")]
    print("step,span_us,busy_us,idle_us,largest_gap_us,largest_gap_after")
    for k, (a, b) in enumerate(zip(idx, idx[1:])):
        seg = rows[a:b]
        busy = sum(e - s for _, s, e in seg)
        span = rows[b][1] - seg[0][1]
        g = [(seg[i + 1][1] - seg[i][2], names[a + i]) for i in range(len(seg) - 1)] + [(rows[b][1] - seg[-1][2], names[b - 1])]
        big = max(g)
        print(f"{k},{span / 1e3:.1f},{busy / 1e3:.1f},{(span - busy) / 1e3:.1f},{big[0] / 1e3:.1f},{big[1]}")


def pmc_csv(paths):
    """rocprofv3 --output-format csv: *_counter_collection.csv files (one row per dispatch and counter)."""
    import csv
    table = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    counters = []
    for path in paths:
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            table[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            # no VGPR columns: rocprofv3's VGPR_Count is an allocation-granule figure (128 for a 253-register kernel, 256 for a
            # 512-register one) — the truth is the code object's metadata, tools/kernel_resources.sh
            meta[k] = (r["Workgroup_Size"], r["LDS_Block_Size"], r["Scratch_Size"])
            if r["Counter_Name"] not in counters:
                counters.append(r["Counter_Name"])
    print("kernel,dispatches,wg_size,lds_bytes,scratch," + ",".join(counters))
    for k, d in sorted(table.items(), key=lambda kv: -sum(kv[1].get(counters[0], [0]))):
        n = max(len(v) for v in d.values())
        print(k + f",{n}," + ",".join(meta[k]) + "," + ",".join(f"{sum(d[c]) / len(d[c]):.4e}" if d.get(c) else "" for c in counters))


def hbm_csv(fetch_path, write_path, batch):
    """Two --pmc passes (FETCH_SIZE, WRITE_SIZE; both in KB) -> per-kernel HBM traffic per launch and per DAG, the launches
    per train step, and a TOTAL row per step.  FETCH_SIZE is doubled: gfx950 counts its 128-byte read requests as 64
    (MI355X_MICROARCH.md, HBM section)."""
    import csv

    def avg(path, counter):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k.startswith("k_") and r["Counter_Name"] == counter:
                acc[k].append(float(r["Counter_Value"]))
        steps = max(len(acc.get("k_pack", [])), len(acc.get("k_pack_w", [])), len(acc.get("k_build_records", [])), 1)
        return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) / steps for k, v in acc.items()}
    (fetch, per_step), (write, _) = avg(fetch_path, "FETCH_SIZE"), avg(write_path, "WRITE_SIZE")
    print("kernel,launches_per_step,fetch_KB_corrected,write_KB,bytes_per_DAG")
    total = 0.0
    for k in sorted(fetch, key=lambda k: -(2 * fetch[k] + write.get(k, 0.0)) * per_step[k]):
        f, w = 2 * fetch[k], write.get(k, 0.0)
        total += (f + w) * 1024 * per_step[k]
        print(f"{k},{per_step[k]:.2f},{f:.0f},{w:.0f},{(f + w) * 1024 / batch:.0f}")
    print(f"TOTAL_PER_STEP,1,,,{total / batch:.0f}")


def sq_derived(path):
    """Derived figures from an SQ counter pass (pmc-csv output): waves parked / issue-stalled / issuing as fractions of
    SQ_WAVE_CYCLES (all quad-cycles), and the matrix pipe's busy share for 512-thread workgroups at one per CU (two waves
    per SIMD share a pipe): SQ_VALU_MFMA_BUSY_CYCLES (cycles) / (4 * SQ_WAVE_CYCLES / 2)."""
    import csv
    rows = list(csv.DictReader(open(path)))
    print("kernel,wait_any_frac,issue_stall_frac,issuing_frac,valu_frac,mfma_pipe_busy,lds_conflict_per_lds_inst")
    for r in rows:
        try:
            wc = float(r["SQ_WAVE_CYCLES"])
            waves_per_simd = max(1.0, float(r["wg_size"]) / 256.0)
            print(f"{r['kernel']},{float(r['SQ_WAIT_ANY']) / wc:.3f},{float(r['SQ_WAIT_INST_ANY']) / wc:.3f},"
                  f"{float(r['SQ_ACTIVE_INST_ANY']) / wc:.3f},{float(r['SQ_ACTIVE_INST_VALU']) / wc:.3f},"
                  f"{float(r['SQ_VALU_MFMA_BUSY_CYCLES']) * waves_per_simd / (4 * wc):.3f},"
                  f"{float(r['SQ_LDS_BANK_CONFLICT']) / max(float(r['SQ_INSTS_LDS']), 1.0):.2f}")
        except (KeyError, ValueError, ZeroDivisionError):
            continue


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        stats(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
    elif mode == "gaps":
        gaps(sys.argv[2])
    elif mode == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 8)
    elif mode == "hbm-csv":
        hbm_csv(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    elif mode == "pmc-csv":
        pmc_csv(sys.argv[2:])
    elif mode == "sq-derived":
        sq_derived(sys.argv[2])
    else:
        raise SystemExit(__doc__)
