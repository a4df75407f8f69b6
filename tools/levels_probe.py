#!/usr/bin/env python3
"""The per-process timing levels of the two BASELINE fk kernels (J = 22 at 2^20 frames, J = 52 at 2^18), named with clocks and counters.

    python tools/levels_probe.py [N]             N fresh processes (default 8), each: fk J=22, its copy kernel, fk J=52, its copy kernel run back to
                                                 back for LP_SECONDS (default 2.5) each while a side thread reads the GPU metrics table (amdsmi: gfx clock
                                                 per XCD, fclk is not in the table -> uclk / socclk / socket power / throttle status) as fast as it can;
                                                 only samples taken while the kernel was running AND with gfx > 1 GHz are kept.  The J = 52 kernel is also
                                                 timed on two further sets of allocations made later in the same process (is the level the process's or
                                                 the allocation's?).
    python tools/levels_probe.py --pmc [N]       N more processes under rocprofv3 --pmc (own passes, --kernel-trace only): memory-side latency / stall /
                                                 translation counters per kernel, each process's level read from the kernel durations of the same run.
    python tools/levels_probe.py --child ...     (internal)

Prints one line per process and a summary (fast / slow medians and the relative difference of every quantity)."""
import ctypes as C
import csv
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SECONDS = float(os.environ.get("LP_SECONDS", "2.5"))
PMC_SETS = [
    "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_REQUEST_sum",
    "TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum GRBM_GUI_ACTIVE",
    "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_GUI_ACTIVE",
    "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE",
    "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE",
    "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum GRBM_GUI_ACTIVE",
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_sum GRBM_GUI_ACTIVE",
    "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum GRBM_GUI_ACTIVE",
]  # (TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_WRITE_REQ_LATENCY_sum: that pass never finished on two boxes -- left out)


# ---------------------------------------------------------------------------------------------------------------- child
class Sampler(threading.Thread):
    """reads the GPU metrics table (one sysfs read of gpu_metrics per call) in a loop; each sample is stamped with the host clock"""

    def __init__(self):
        super().__init__(daemon=True)
        self.samples, self.stop_flag, self.err = [], False, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            self.amdsmi, self.h = amdsmi, amdsmi.amdsmi_get_processor_handles()[0]
        except Exception as ex:  # noqa: BLE001
            self.amdsmi, self.err = None, repr(ex)[:200]

    def run(self):
        if self.amdsmi is None: return
        while not self.stop_flag:
            t = time.perf_counter()
            try:
                m = self.amdsmi.amdsmi_get_gpu_metrics_info(self.h)
            except Exception as ex:  # noqa: BLE001
                self.err = repr(ex)[:200]; return
            self.samples.append((t, m))
            time.sleep(0.004)

    def window(self, t0, t1):
        """per quantity the median over the samples of [t0, t1] with every reported XCD clock above 1 GHz"""
        def nums(v):
            if isinstance(v, (list, tuple)): return [x for x in v if isinstance(x, (int, float)) and 0 < x < 60000]
            return [v] if isinstance(v, (int, float)) and 0 < v < 10 ** 9 else []
        kept = []
        for t, m in self.samples:
            if not (t0 <= t <= t1): continue
            g = nums(m.get("current_gfxclks", m.get("current_gfxclk")))
            if not g or min(g) <= 1000: continue
            kept.append(m)
        out = {"n": len(kept)}
        if not kept: return out
        keys = ("current_gfxclks", "current_socclks", "current_uclk", "current_vclks", "current_dclks", "current_socket_power", "average_socket_power",
                "temperature_hotspot", "temperature_mem", "average_gfx_activity", "average_umc_activity", "throttle_status", "indep_throttle_status",
                "gfxclk_lock_status", "current_fclk", "average_fclk_frequency", "average_uclk_frequency", "average_gfxclk_frequency",
                "average_socclk_frequency", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc",
                "hbm_thm_residency_acc", "pcie_bandwidth_inst")
        for k in keys:
            vals = [x for m in kept for x in nums(m.get(k))]
            if vals:
                out[k] = statistics.median(vals)
                if k == "current_gfxclks":
                    out["gfx_min"], out["gfx_max"] = min(statistics.median(nums(m[k])) for m in kept), max(statistics.median(nums(m[k])) for m in kept)
                if k.endswith("_acc"): out[k] = max(vals) - min(vals)  # residency counters: what accumulated inside the window
        return out


def child(pmc_mode):
    import numpy as np
    import torch
    from pymotion_amd import _lib
    from pymotion_amd import synthetic as syn
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def workload(J, F, par):
        par = np.ascontiguousarray(par, dtype=np.int32)
        src = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
        off = torch.randn((J, 3), device="cuda") * 0.1; off[0] = 0
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
        big = torch.empty((F, J, 12), device="cuda")
        fk = lambda: _lib.call("pm_fk_f32", P(src), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)  # noqa: E731
        cp = lambda: _lib.call("pm_stream_ceiling_f32", P(src), P(big), F, 4 * J, 12 * J, None)  # noqa: E731
        return fk, cp, (src, root, off, pos, rm, big, par)

    def sustained(fn, seconds, sampler=None):
        for _ in range(30): fn()
        torch.cuda.synchronize()
        # how many launches fill `seconds`
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        n = max(50, int(seconds * 1e3 / (e0.elapsed_time(e1) / 50)))
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n): fn()
        e1.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        us = e0.elapsed_time(e1) / n * 1e3
        # the queue drains at the kernel's own rate: the GPU was busy from t0 (+ a launch) to t1; keep the middle 80 %
        w = sampler.window(t0 + 0.1 * (t1 - t0), t1 - 0.1 * (t1 - t0)) if sampler else None
        return us, w

    if pmc_mode:
        # (one warm-up workload first so that the three J = 52 sets are allocations made "later", like the slow ones of the timing mode)
        fk, cp, keep22 = workload(22, 1 << 20, syn.PARENTS_22)
        for _ in range(5): fk()
        sets = [workload(52, 1 << 18, syn.PARENTS_52) for _ in range(3)]
        for rep in range(2):
            for fk, cp, keep in sets:
                for _ in range(20): fk()
                torch.cuda.synchronize()
        print("PMCSETS " + json.dumps([[hex(t.data_ptr()) for t in keep[:5] if hasattr(t, "data_ptr")] for _, _, keep in sets]), flush=True)
        return
    smp = Sampler(); smp.start()
    rec = {"pid": os.getpid()}
    try:
        import amdsmi
        rec["uuid"] = amdsmi.amdsmi_get_gpu_device_uuid(smp.h)[-12:]
    except Exception:  # noqa: BLE001
        pass
    keepalive = []
    for J, F, par in ((22, 1 << 20, syn.PARENTS_22), (52, 1 << 18, syn.PARENTS_52)):
        fk, cp, keep = workload(J, F, par)
        keepalive.append(keep)
        us, w = sustained(fk, SECONDS, smp)
        rec[f"fk{J}_us"], rec[f"fk{J}_clk"] = round(us, 2), w
        us, w = sustained(cp, SECONDS, smp)
        rec[f"copy{J}_us"], rec[f"copy{J}_clk"] = round(us, 2), w
    # the J = 52 kernel on allocations made later in the same process (earlier ones stay alive), then every set AGAIN in the same order:
    # is the level the allocation's (it comes back with the set) or the moment's (it follows the clock / the time under load)?
    sets = [(fk, keep)]
    for _ in range(2):
        fk2, cp2, keep2 = workload(52, 1 << 18, syn.PARENTS_52)
        keepalive.append(keep2)
        sets.append((fk2, keep2))
    visits = []
    for rep in range(3):
        for si, (f, keep) in enumerate(sets):
            us, w = sustained(f, 0.8, smp)
            visits.append({"set": si, "us": round(us, 2), "gfx": w.get("current_gfxclks"), "W": w.get("current_socket_power"), "n": w.get("n"),
                           "Thot": w.get("temperature_hotspot"), "Tmem": w.get("temperature_mem")})
    rec["fk52_visits"] = visits
    rec["fk52_us_later_allocations"] = [v["us"] for v in visits[1:3]]
    rec["fk52_set_pointers"] = [[hex(t.data_ptr()) for t in keep[:5] if hasattr(t, "data_ptr")] for _, keep in sets]
    smp.stop_flag = True
    rec["sampler_error"] = smp.err
    rec["samples"] = len(smp.samples)
    print("LEVELS " + json.dumps(rec), flush=True)


# --------------------------------------------------------------------------------------------------------------- parent
def run_children(n):
    recs = []
    for i in range(n):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, timeout=600)
        line = next((ln for ln in r.stdout.splitlines() if ln.startswith("LEVELS ")), None)
        if line is None:
            print(f"process {i}: no record (rc {r.returncode}) {r.stderr[-400:]}", flush=True)
            continue
        rec = json.loads(line[7:]); recs.append(rec)
        print("RAW " + line[7:], flush=True)
        c22, c52 = rec.get("fk22_clk") or {}, rec.get("fk52_clk") or {}
        print(f"process {i}: fk22 {rec['fk22_us']:.1f} us (copy {rec['copy22_us']:.1f}) gfx {c22.get('current_gfxclks')} uclk {c22.get('current_uclk')} soc "
              f"{c22.get('current_socclks')} W {c22.get('current_socket_power')} n {c22.get('n')} | fk52 {rec['fk52_us']:.1f} us (copy {rec['copy52_us']:.1f}; later "
              f"allocations {rec['fk52_us_later_allocations']}) gfx {c52.get('current_gfxclks')} uclk {c52.get('current_uclk')} soc {c52.get('current_socclks')} "
              f"W {c52.get('current_socket_power')} n {c52.get('n')}", flush=True)
        print("   fk52 by allocation set, visited in turn: " + "  ".join(f"set{v['set']} {v['us']:.1f}us gfx {v['gfx']} W {v['W']} Tmem {v['Tmem']}" for v in rec.get("fk52_visits", [])), flush=True)
        print("   pointers (src, root, off, pos, rotmats) per set: " + str(rec.get("fk52_set_pointers")), flush=True)
        time.sleep(1.0)
    return recs


def summarise(recs):
    print("\n# summary: per kernel, processes split at the midpoint of the fastest and slowest launch time; medians per side, slow / fast - 1")
    for k in ("fk22", "fk52"):
        us = [r[f"{k}_us"] for r in recs]
        if len(us) < 2: continue
        lo, hi = min(us), max(us)
        print(f"## {k}: {sorted(us)}  spread {100 * (hi / lo - 1):.1f} %")
        if hi / lo < 1.03:
            print("   (one level in this call)"); continue
        mid = 0.5 * (lo + hi)
        fast, slow = [r for r in recs if r[f"{k}_us"] <= mid], [r for r in recs if r[f"{k}_us"] > mid]
        keys = sorted({q for r in recs for q in (r.get(f"{k}_clk") or {})})
        print(f"   {'quantity':34s} {'fast (n=%d)' % len(fast):>14s} {'slow (n=%d)' % len(slow):>14s}   slow/fast-1")
        def med(rs, f):
            v = [f(r) for r in rs if f(r) is not None]
            return statistics.median(v) if v else None
        rows = [("launch us", lambda r: r[f"{k}_us"]), ("copy kernel us", lambda r: r[f"copy{k[2:]}_us"])]
        rows += [(q, (lambda q: lambda r: (r.get(f"{k}_clk") or {}).get(q))(q)) for q in keys]
        rows += [("copy: " + q, (lambda q: lambda r: (r.get(f"copy{k[2:]}_clk") or {}).get(q))(q)) for q in ("current_gfxclks", "current_uclk", "current_socclks", "current_socket_power")]
        for name, f in rows:
            a, b = med(fast, f), med(slow, f)
            if a is None or b is None: continue
            rel = f"{100 * (b / a - 1):+7.1f} %" if a else "      -"
            print(f"   {name:34s} {a:14.2f} {b:14.2f}   {rel}")


def run_pmc(n):
    os.makedirs("/tmp/lp", exist_ok=True)
    table = []
    for i in range(n):
        cset = PMC_SETS[i % len(PMC_SETS)]
        d = f"/tmp/lp/p{i}"
        subprocess.run(["rm", "-rf", d])
        env = dict(os.environ, PYTHONPATH=ROOT, TMPDIR="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--kernel-include-regex", "pm::", "--pmc", *cset.split(), "-d", d, "-o", "q", "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__), "--child", "--pmc"]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get("LP_PMC_TIMEOUT", "150")), cwd="/tmp", env=env)
        except subprocess.TimeoutExpired:
            print(f"pmc process {i} ({cset.split()[0]} ...): timed out", flush=True); continue
        rows, tr = [], {}
        for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for row in csv.DictReader(open(path, newline="")):
                if "fk_pipe_kernel" in row["Kernel_Name"]:
                    tr[row.get("Dispatch_Id")] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
        cnt = {}
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(path, newline="")):
                if "fk_pipe_kernel" in row["Kernel_Name"]:
                    cnt.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
        ids = sorted(cnt)
        if len(ids) < 120:
            print(f"pmc process {i} ({cset.split()[0]} ...): {len(ids)} dispatches collected (rc {r.returncode}) {r.stderr[-200:]}", flush=True); continue
        ids = ids[-120:]  # three sets x 20 launches, twice
        for seg in range(6):
            seg_ids = ids[20 * seg + 5:20 * seg + 20]  # (the first launches of a visit run on the clocks of the one before)
            rec = {"proc": i, "kernel": f"fk52 set{seg % 3} visit{seg // 3}", "us": round(statistics.median(tr.get(str(x), float("nan")) for x in seg_ids), 2)}
            for c in cnt[seg_ids[0]]:
                rec[c] = statistics.median(cnt[x].get(c, float("nan")) for x in seg_ids)
            table.append(rec)
            print("PMC " + json.dumps(rec), flush=True)
    # derived per-request figures
    print("\n# per kernel and process: duration under the profiler, then counters (medians over the dispatches) and counter / request ratios")
    for rec in table:
        d = dict(rec)
        def ratio(a, b):
            return round(d[a] / d[b], 2) if a in d and b in d and d[b] else None
        extra = {"rd_latency_cycles_per_req": ratio("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"),
                 "wr_latency_cycles_per_req": ratio("TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_WRITE_REQ_sum"),
                 "utcl1_miss_per_request": ratio("TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_REQUEST_sum"),
                 "ea_rd_level_per_req": ratio("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum"),
                 "ea_wr_level_per_req": ratio("TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum"),
                 "rd_dram_fraction": ratio("TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_sum"), "wr_dram_fraction": ratio("TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_WRREQ_sum"),
                 "tcc_hit_rate": ratio("TCC_HIT_sum", "TCC_REQ_sum")}
        extra["utcl2_busy_fraction"] = ratio("GRBM_UTCL2_BUSY", "GRBM_GUI_ACTIVE")
        print(rec["proc"], rec["kernel"], rec["us"], {k: v for k, v in extra.items() if v is not None})


def short(name, grid):
    name = name.replace("void ", "")
    return name[:name.find("(")] if "(" in name else name[:90]


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--child":
        child("--pmc" in args)
    elif args and args[0] == "--pmc":
        run_pmc(int(args[1]) if len(args) > 1 else 10)
    else:
        recs = run_children(int(args[0]) if args else 8)
        summarise(recs)
