for cfg in "40 96=2" "0 96=2" "40 96=1"; do set -- $cfg; echo "== occ3 $1 force $2"; NQ_IG3_OCC3_STEPS=$1 NQ_MI3_FORCE=$2 python tools/bench_nerv_tail.py 2>&1 | grep -v amdgpu.ids; done
