export RSBA_BENCH_TEST_ONE_GPU=1
for n in 2 4 8; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 10 --warmup 2 > gpurun_out/bench_hook_c4_n$n.json 2> gpurun_out/bench_hook_c4_n$n.err
  echo "n=$n rc=$?"
done
for n in 4 8; do
  timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2952$n bench.py --config C5 --lm-iters 6 --gpus $n --steps 5 --warmup 1 > gpurun_out/bench_hook_c5_n$n.json 2> gpurun_out/bench_hook_c5_n$n.err
  echo "c5 n=$n rc=$?"
done
