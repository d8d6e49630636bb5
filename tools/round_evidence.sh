# the whole evidence run of a round on the GPU box: bench lines + kernel stats + PMC passes, the text evidence, the multi-rank hook lines
export TMPDIR=/tmp
R=${1:-r06}
bash tools/profile_round.sh $R > gpurun_out/$R.round.log 2>&1
bash tools/profile_texts.sh $R > gpurun_out/$R.texts.log 2>&1
O=gpurun_out/$R
for N in 2 4 8; do
  RSBA_BENCH_TEST_ONE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 10 --warmup 2 --lm-iters 8 > $O/bench_hook_c4_n$N.json 2> $O/bench_hook_c4_n$N.err
done
RSBA_BENCH_TEST_ONE_GPU=1 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --config C5 --steps 10 --warmup 2 --lm-iters 6 > $O/bench_hook_c5_n8.json 2> $O/bench_hook_c5_n8.err
# the branch of bench.py the driver's multi-GPU command takes (attach_rccl -> rsba_rccl_comm_create -> rsba_set_exchange_rccl, the LM leg under its watchdog), over the stream-ordered stand-in for librccl
for N in 2 8; do
  RSBA_BENCH_TEST_ONE_GPU=1 RSBA_BENCH_NATIVE=1 RSBA_RCCL_LIB=tools/libmock_rccl.so timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29520 + N)) bench.py --gpus $N --steps 10 --warmup 2 --lm-iters 8 > $O/bench_native_mock_c4_n$N.json 2> $O/bench_native_mock_c4_n$N.err
done
python tools/hook_summary.py $O/bench_hook_c4_n2.json $O/bench_hook_c4_n4.json $O/bench_hook_c4_n8.json $O/bench_hook_c5_n8.json > $O/sharded_per_rank_device_time.txt 2>&1
tail -c 1200 $O/bench.json; cat $O/sharded_per_rank_device_time.txt; cat $O/pytest_gpu.txt | tail -3
