#!/usr/bin/env python3
"""the two timing modes of the J = 52 kernels (0.65 / 0.70 of the HBM spec, by box / process): are they the chip's CLOCK?  The J = 52 walks are
VALU-issue-bound, the 22-joint body is not.  Runs each kernel back to back for a few seconds and samples the shader / memory clocks,
power and temperature from sysfs (and rocm-smi once) WHILE it runs; prints the kernel's average time beside them."""
import ctypes as C, glob, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pymotion_amd import _lib
from pymotion_amd import synthetic as syn
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731


POLL = os.environ.get("FKC_SYSFS") == "1"
NOSMI = os.environ.get("FKC_NOSMI") == "1"


def smi():
    """shader clock (MHz) and socket power (W) as rocm-smi reports them right now (sysfs' hwmon files belong to the host's first card, not
    necessarily the visible one)"""
    if NOSMI: return {}
    try:
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
    except Exception as ex:  # noqa: BLE001
        return repr(ex)
    got = {}
    for ln in out.splitlines():
        if "sclk clock level" in ln: got["sclk"] = ln.split("(")[-1].rstrip(")")
        if "Package Power" in ln: got["W"] = ln.split(":")[-1].strip()
        if "Sensor junction" in ln: got["Tj"] = ln.split(":")[-1].strip()
        if "Sensor memory" in ln: got["Tmem"] = ln.split(":")[-1].strip()
    return got


def amd_smi_clocks():
    """per-XCD shader clocks (amd-smi metric --clock: GFX_0 ... GFX_7), as text"""
    try:
        out = subprocess.run(["/opt/rocm/bin/amd-smi", "metric", "--clock", "--json"], capture_output=True, text=True, timeout=30).stdout
        import json
        j = json.loads(out)
        rec = j[0] if isinstance(j, list) else j
        if isinstance(rec, dict) and "gpu_data" in rec: rec = rec["gpu_data"][0]
        clk = rec.get("clock", rec)
        got = []
        for k in sorted(clk):
            if k.lower().startswith("gfx"):
                v = clk[k]
                c = v.get("clk", v) if isinstance(v, dict) else v
                got.append(f"{k}={c.get('value', c) if isinstance(c, dict) else c}")
        return " ".join(got) or out[:600]
    except Exception as ex:  # noqa: BLE001
        return repr(ex)[:300]


def sysfs_poll():
    """what the first version of this probe did every 1000 launches: read every card's DPM tables and hwmon files (each read is a message to
    the power-management firmware)"""
    n = 0
    for dev in glob.glob("/sys/class/drm/card*/device"):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "gpu_busy_percent"):
            try:
                with open(os.path.join(dev, name)) as f: f.read(); n += 1
            except OSError: pass
        for hw in glob.glob(os.path.join(dev, "hwmon/hwmon*")):
            for name in ("power1_average", "power1_input", "power1_cap", "temp1_input", "temp2_input", "temp3_input", "freq1_input", "freq2_input"):
                try:
                    with open(os.path.join(hw, name)) as f: f.read(); n += 1
                except OSError: pass
    return n


def main():
    print("idle:", smi(), flush=True)
    if POLL: print("sysfs files read per poll:", sysfs_poll())
    try:
        if NOSMI: raise RuntimeError("no rocm-smi asked")
        out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showuniqueid", "--showbus", "--showcomputepartition", "--showmemorypartition", "--showperflevel",
                              "--showmaxpower", "--showclocks"], capture_output=True, text=True, timeout=30).stdout
        print("\n".join(ln for ln in out.splitlines() if ln.startswith("GPU[")), flush=True)
    except Exception as ex:  # noqa: BLE001
        print(repr(ex))
    for J, F, par in ((52, 1 << 18, syn.PARENTS_52), (22, 1 << 20, syn.PARENTS_22), (52, 1 << 21, syn.PARENTS_52), (52, 1 << 15, syn.PARENTS_52))[slice(0, 1) if os.environ.get('FKC_ONLY52') == '1' else slice(0, 4, 2) if os.environ.get('FKC_SIZES') == '1' else slice(0, 2)]:
        par = np.ascontiguousarray(par, dtype=np.int32)
        src = torch.randn((F, J, 4), device="cuda"); root = torch.rand((F, 3), device="cuda") * 4 - 2
        off = torch.randn((J, 3), device="cuda") * 0.1
        pos = torch.empty((F, J, 3), device="cuda"); rm = torch.empty((F, J, 3, 3), device="cuda")
        fn = lambda: _lib.call("pm_fk_f32", P(src), P(root), P(off), 0, par.ctypes.data_as(C.c_void_p), F, J, P(pos), P(rm), None)  # noqa: E731
        for _ in range(50): fn()
        torch.cuda.synchronize()
        for rep in range(2):
            n = (20000 if os.environ.get('FKC_XCD') == '1' else 8000) if F <= (1 << 18) else 1000
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                fn()
                if POLL and i % 1000 == 999: sysfs_poll()
            e1.record()
            xcd = amd_smi_clocks() if os.environ.get("FKC_XCD") == "1" else None
            seen = [smi(), smi()]  # (the queue holds seconds of launches: both samples fall inside the run -- `busy` below says so)
            busy = not e1.query()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            print(f"J={J} F={F}: {us:7.1f} us a launch, {F * (64 * J + 12) / us / 1e3 / 80:5.1f} % | still running at the 2nd sample: {busy} | {seen}", flush=True)
            if xcd: print("   per-XCD clocks while it ran:", xcd, flush=True)
            time.sleep(0.2 if NOSMI else 1.0)
        del src, pos, rm


if __name__ == "__main__":
    main()
