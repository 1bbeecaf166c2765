#!/bin/bash
# Land tools/r5_patches in the product tree (run in the container, then the full GPU suite + bench on the MI355X before committing):
#   bash tools/r5_land_patches.sh && gpurun --timeout 1500 -- 'python -m pytest tests -x -q -m gpu; python -c "import __graft_entry__ as g; g.smoke()"; python bench.py'
set -e
cd "$(dirname "$0")/.."
# 0006 (the long-K loop) is optional: land it with `bash tools/r5_land_patches.sh --with-0006` once its probe says so
PATCHES=$(ls tools/r5_patches/000[1-5]*.patch)
if [ "$1" = "--with-0006" ]; then PATCHES="$PATCHES $(ls tools/r5_patches/0006*.patch)"; fi
for p in $PATCHES; do
  git apply --check "$p"
  git apply "$p"
  echo "applied $p"
done
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests -x -q -m "not gpu"
