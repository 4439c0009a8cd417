cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/flaky2
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/flaky2/run$i.log 2>&1; tail -1 gpurun_out/flaky2/run$i.log; grep "^FAILED" gpurun_out/flaky2/run$i.log; done
