set -e
for nb in 0 1; do
for blk in "2 2 2" "4 4 4" "8 8 8" "4 4 2" "8 8 2" "16 16 4"; do
  echo "== class-major block $blk numbering $nb"
  BP5_BRICK_ORDER=1 timeout -k 10 120 python tools/bench_apply.py --cell-block $blk --numbering $nb --variants 0 --rounds 3 2>/dev/null | grep variant
done; done
echo "== lexicographic in brick, 4 4 4, numbering 1"
timeout -k 10 120 python tools/bench_apply.py --cell-block 4 4 4 --numbering 1 --variants 0 --rounds 3 2>/dev/null | grep variant
echo "== plain lexicographic mesh"
timeout -k 10 120 python tools/bench_apply.py --variants 0 --rounds 3 2>/dev/null | grep variant
