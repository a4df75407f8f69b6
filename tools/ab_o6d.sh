for i in 1 2; do
echo "== old (ab)"; PMHIP_VARIANT=ab python tools/perf_probe.py --only o6d --sustained 100 2>&1 | grep -v amdgpu.ids
echo "== new (prod)"; python tools/perf_probe.py --only o6d --sustained 100 2>&1 | grep -v amdgpu.ids
done
