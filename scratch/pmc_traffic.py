#!/usr/bin/env python3
"""Merges the rocprofv3 --pmc passes of one workload (scratch/pmc_all.sh <tag> ...) into profiles/round3/hbm_traffic.json[<bench tag>].

  python scratch/pmc_traffic.py <pass tag> <bench tag> [limiter text]

Memory-side read bytes of a kernel = 128 x TCC_EA0_RDREQ_128B_sum + 64 x TCC_EA0_RDREQ_64B_sum + 32 x TCC_EA0_RDREQ_32B_sum: the
requests L2 sends to the fabric, by size (on gfx950 FETCH_SIZE tallies the 128-byte requests at 64 bytes, so 2 x FETCH_SIZE is the
same number: MI355X_MICROARCH.md, HBM; checked on patterns of known byte count by scratch/ubench/hbm_calib.hip,
profiles/round2/hbm_counter_calibration.txt).  Write bytes = WRITE_SIZE.  Infinity-Cache hits are included: this is the traffic
beyond L2, an upper bound of the HBM traffic."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, btag = sys.argv[1], sys.argv[2]
limiter = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] else None
family = sys.argv[4] if len(sys.argv) > 4 else "march_kernel"      # kernel family the counters are summed over (configs[1]: event_kernel runs the walks)


def sums(kind):
    """counter sums of K_march from the pass's summary.txt (written on the GPU box by pmc_sum.py; the raw CSVs above 1 MiB are not kept)"""
    import re
    d = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, kind))
    out, cur = {}, None
    for line in open(os.path.join(d, "summary.txt")):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        m = re.match(r"\s+(\S+)\s+(\S+)\s+\(dispatches (\d+)\)", line)
        if m and cur == family:
            out[m.group(1)] = {"sum": float(m.group(2)), "dispatches": int(m.group(3))}
    return out, d


rd, d_rd = sums("rd"); fe, _ = sums("fetch"); wr, _ = sums("write"); l2, _ = sums("l2"); sq, _ = sums("sq"); sq2, _ = sums("sq2")
bench = json.loads(open(os.path.join(d_rd, "stdout.txt")).read().strip().splitlines()[-1])
c = bench["counters_per_step"]
steps = c["eikonal_steps"] or c["tentative_collisions"]; wave_steps = c["lane_slots"] / 64.0
v = lambda d, k: d.get(k, {}).get("sum", 0.0)
read_b = 128 * v(rd, "TCC_EA0_RDREQ_128B_sum") + 64 * v(rd, "TCC_EA0_RDREQ_64B_sum") + 32 * v(rd, "TCC_EA0_RDREQ_32B_sum")
write_b = 1024 * v(wr, "WRITE_SIZE")
sys.path.insert(0, ROOT)
import bench as _bench
entry = {
    # what the pass was taken on: bench.py uses a committed entry only for the same kernel sources, RIF layout and options
    "source_hash": _bench.source_hash(), "layout": bench["config"]["layout"], "options": "pipes=1",
    "kernel_family": family,
    "method": "rocprofv3 --pmc, separate passes of one single-pipeline bench step each (scratch/pmc_all.sh %s): read = 128*RDREQ_128B + 64*RDREQ_64B + 32*RDREQ_32B, write = WRITE_SIZE; includes Infinity-Cache hits" % tag,
    "workload": bench["config"]["workload"], "march_launches": rd.get("TCC_EA0_RDREQ_sum", {}).get("dispatches"),
    "eikonal_steps": steps, "wave_steps": wave_steps,
    "read_bytes": read_b, "write_bytes": write_b, "two_x_FETCH_SIZE_bytes": 2 * 1024 * v(fe, "FETCH_SIZE"),
    "hbm_bytes_per_eikonal_step": (read_b + write_b) / steps,
    "rdreq": {k: v(rd, k) for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_32B_sum")},
    "l2": {"TCC_REQ": v(l2, "TCC_REQ_sum"), "TCC_HIT": v(l2, "TCC_HIT_sum"), "TCC_MISS": v(l2, "TCC_MISS_sum"),
           "hit_rate": v(l2, "TCC_HIT_sum") / max(v(l2, "TCC_HIT_sum") + v(l2, "TCC_MISS_sum"), 1.0),
           "l2_requests_per_eikonal_step": v(l2, "TCC_REQ_sum") / steps},
    "sq": {k: v(sq, k) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY")},
    "sq2": {k: v(sq2, k) for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "GRBM_GUI_ACTIVE")},
    "valu_inst_per_wave_step": v(sq, "SQ_INSTS_VALU") / max(wave_steps, 1.0),
    "wave_time_split": {"waiting_on_memory_or_barrier (SQ_WAIT_ANY)": v(sq2, "SQ_WAIT_ANY") / max(v(sq, "SQ_WAVE_CYCLES"), 1.0),
                        "issue_stall (SQ_WAIT_INST_ANY)": v(sq, "SQ_WAIT_INST_ANY") / max(v(sq, "SQ_WAVE_CYCLES"), 1.0),
                        "issuing (SQ_ACTIVE_INST_ANY)": v(sq2, "SQ_ACTIVE_INST_ANY") / max(v(sq, "SQ_WAVE_CYCLES"), 1.0)},
}
if limiter:
    entry["limiter"] = limiter
path = os.path.join(ROOT, "profiles", "round3", "hbm_traffic.json")
allv = json.load(open(path)) if os.path.exists(path) else {}
allv[btag] = entry
json.dump(allv, open(path, "w"), indent=1)
print(btag, "hbm bytes / eikonal step %.1f, L2 hit rate %.3f, VALU / wave-step %.0f" % (entry["hbm_bytes_per_eikonal_step"], entry["l2"]["hit_rate"], entry["valu_inst_per_wave_step"]))
