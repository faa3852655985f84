# usage: bash tools/prof_dp8_one_gpu.sh  -> kernel traces of an 8-rank data-parallel run with all ranks on ONE GPU (gloo):
# the device-side cost of the N = 8 step (merge of 8 row lists, optimizer over 8 x the entries), not its communication
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
rm -rf $O/prof_dp8
SWR_BENCH_BACKEND=gloo timeout 900 rocprofv3 --kernel-trace -d $O/prof_dp8 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/prof_dp8.log 2>&1
grep "timed region" $O/prof_dp8.log; ls $O/prof_dp8/*/ | head -20
