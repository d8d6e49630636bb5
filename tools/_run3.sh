O=gpurun_out/r02; mkdir -p $O
python -m pytest tests/test_gpu_fullsize.py -m gpu -q > $O/gputest3.log 2>&1; echo "pytest rc $?" >> $O/gputest3.log
tail -30 $O/gputest3.log
bash tools/profile_round.sh r02 > $O/profile_round.log 2>&1
tail -5 $O/profile_round.log | head -c 3000
