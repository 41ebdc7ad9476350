cd $GRAFT_REPO_ROOT; O=gpurun_out/r5h; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s > $O/model.log 2>&1; echo "model rc=$?" | tee -a $O/summary.txt
timeout 1800 python -m pytest tests/test_fullsize_gpu.py -m gpu -x -q -k "cfg3_shape or cfg2_shape or ragged or bn8k_fixture" > $O/fullsize.log 2>&1; echo "fullsize rc=$?" | tee -a $O/summary.txt
grep -n "^trajectory\|^\.*trajectory" $O/model.log | cut -c1-3000; tail -n 3 $O/model.log; tail -n 3 $O/fullsize.log; cat $O/summary.txt
