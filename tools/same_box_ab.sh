#!/bin/bash
# Same-box A/B of two trees: the tree at tools/_variants/base_tree (an exported earlier commit, built there) and this one, benched
# alternately on ONE box (boxes differ by several %):   tools/same_box_ab.sh [rounds]
rounds=${1:-3}
here=$(pwd)
for r in $(seq 1 $rounds); do
  for t in tools/_variants/base_tree .; do
    v=$(cd $t && python bench.py --no-extra --no-roofline --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | python -c "import sys, json; print(json.loads([l for l in sys.stdin if l.startswith('{')][0])['value'])")
    echo "round $r $( [ $t = . ] && echo this-tree || echo base-tree ): $v frames/s"
  done
done
