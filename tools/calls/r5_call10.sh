cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_fz; mkdir -p $O
for c in english mixed code_files; do
TD_HIP_LIB=variants/fz_timing.so TD_BENCH_GRAPH=0 timeout 300 python bench.py --corpus $c --size-mb 256 --no-cpu-baseline --no-verify --steps 1 --warmup 0 2>&1 | grep "fused wg" | head -400 > $O/fz_$c.txt
python - $O/fz_$c.txt $c <<'PY'
import sys, re
names = "stage masks rules heads list probe cold bookkeeping lookback docs tail place t12 t13".split()
tot = [0] * 14; total = 0; tiles = 0; n = 0
for line in open(sys.argv[1]):
    m = re.search(r"(\d+) tiles, total (\d+) cycles \| stage (\d+) masks (\d+) rules (\d+) heads (\d+) \| list (\d+) probe (\d+) cold (\d+) bookkeeping (\d+) lookback (\d+) docs (\d+) tail (\d+) place (\d+) \| t12 (\d+) t13 (\d+)", line)
    if not m: continue
    v = list(map(int, m.groups())); tiles += v[0]; total += v[1]; n += 1
    for i in range(14): tot[i] += v[2 + i]
print(sys.argv[2], n, "workgroups sampled,", tiles, "tiles, cycles per tile", total // max(tiles, 1), " ".join(f"{nm} {100 * t / max(total, 1):.1f}%" for nm, t in zip(names, tot)))
PY
done
