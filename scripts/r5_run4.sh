cd $GRAFT_REPO_ROOT; O=gpurun_out/r5d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -s -k "round5" > $O/model.log 2>&1; echo "model tests rc=$?" | tee -a $O/summary.txt
tail -n 12 $O/model.log; cat $O/summary.txt
