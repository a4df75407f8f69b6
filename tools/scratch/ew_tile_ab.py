import subprocess, os, re, sys, collections
res = collections.OrderedDict()
for rnd in range(2):
    for v in ("ab", "prod", "ab2"):
        env = dict(os.environ, PMHIP_VARIANT=v)
        out = subprocess.run([sys.executable, "tools/perf_probe.py", "--only", "ew", "--sustained", "60"], env=env, capture_output=True, text=True).stdout
        for l in out.splitlines():
            m = re.match(r"(\S.*?)\s+([\d.]+) us \(min", l)
            if m: res.setdefault(m.group(1), {}).setdefault(v, []).append(float(m.group(2)))
print(f"{'op':34s} {'64/wave':>16s} {'128/wave':>16s} {'256/wave':>16s}")
for k, d in res.items():
    print(f"{k:34s} " + " ".join(f"{'/'.join('%.1f' % x for x in d.get(v, [])):>16s}" for v in ("ab", "prod", "ab2")))
