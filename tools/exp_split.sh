# the two split workloads on one GPU, and the N = 2 control flow of all three workloads rehearsed on one GPU over gloo
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --workload esa4k --steps 3 --warmup 1 > gpurun_out/split_esa4k.json
timeout -k 10 300 python bench.py --workload la8k --steps 3 --warmup 1 > gpurun_out/split_la8k.json
for wl in me1080p esa4k la8k; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --workload $wl --rehearse-one-gpu --cpu-pairs 2 --ref-pairs 0 --no-other-paths --pairs 8 > gpurun_out/rehearse_$wl.json 2> gpurun_out/rehearse_$wl.err
done
for f in gpurun_out/split_*.json gpurun_out/rehearse_*.json; do echo $f; cut -c1-260 $f; done
