cd $GRAFT_REPO_ROOT; O=gpurun_out/r5l; mkdir -p $O
export TMPDIR=/tmp
for v in 0 1; do MC_LANE_XMAP=$v timeout 600 python scripts/dw_form_ab.py > $O/dw_xmap$v.txt 2>&1; done
paste -d'\n' $O/dw_xmap0.txt $O/dw_xmap1.txt | grep "c=  240\|c=  384\|c= 1824\|c= 3072\|c=  768"
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "dwconv" 2>&1 | tail -2
