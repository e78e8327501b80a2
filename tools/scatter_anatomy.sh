# scatter anatomy: full kernel / walk without atomics / cells + sort only (ESLAM_SC_MODE 0 / 1 / 2), slab reduction off
for cfg in "${@:-room0 4096 56 8}"; do
 for m in 0 1 2 3 4 5 6; do ESLAM_SC_MODE=$m ESLAM_SC_NO_REDUCE=1 python tools/dbg_scatter.py $cfg 2>/dev/null | cut -c1-100; done
done
