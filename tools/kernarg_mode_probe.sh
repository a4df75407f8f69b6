# the per-process timing levels of the J = 52 kernels: do they come with the address-space layout?  processes alternate between the default
# (randomised mmap base: torch's allocations, the code object and the kernel-argument pool land elsewhere each time) and `setarch -R`
export FKC_NOSMI=1 FKC_ONLY52=1
setarch x86_64 -R true && echo "setarch -R works" || echo "setarch -R refused"
for i in 1 2 3 4 5 6; do
echo "## process $i default"; timeout 120 python tools/fk_clock_probe.py 2>&1 | grep "J="
echo "## process $i setarch -R"; setarch x86_64 -R timeout 120 python tools/fk_clock_probe.py 2>&1 | grep "J="
done
