: > gpurun_out/sk_sizes.jsonl
for n in 512 1024 1536 2048 3072 4096; do python tools/bench_models.py sk $n 2048 2>/dev/null | tail -1 >> gpurun_out/sk_sizes.jsonl; done
./tools/ubench/skh_bench_stamps.out h8 1024 2048 65536 > gpurun_out/skh_stamps.txt 2>&1
./tools/ubench/skh_bench_stamps.out h8 4096 2048 32768 > gpurun_out/skh_stamps_4096.txt 2>&1
bash tools/profile_model.sh c3 sk 1024 2048 > gpurun_out/prof_c3.log 2>&1
cat gpurun_out/sk_sizes.jsonl | cut -c1-200
