cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/abl
for ab in 0 1 2 4 7; do
  ADT_SEQ_ABLATE=$ab timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl/a$ab -o p -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/abl/log$ab.txt 2>&1 || exit 1
done
