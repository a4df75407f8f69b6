# the per-process timing levels of the J = 52 kernels: kernel or launch gap?  each process times 2^18 frames (8000 launches) and 2^21 frames (1000 launches)
export FKC_NOSMI=1 FKC_SIZES=1
for i in 1 2 3 4 5 6 7 8; do echo "## process $i"; timeout 120 python tools/fk_clock_probe.py 2>&1 | grep "J="; done
