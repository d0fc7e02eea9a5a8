# Round-end numbers (run on the GPU box through gpurun): the default line (headline + extra_configs) and the other levels.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/r03_bench_default.json 2> $O/default.err
python bench.py --input text --level 3 --no-cpu-baseline --no-extra > $O/r03_bench_text_L3.json 2> $O/text3.err
python bench.py --input text --level 5 --no-cpu-baseline --no-extra > $O/r03_bench_text_L5.json 2> $O/text5.err
python bench.py --input text --level 9 --no-cpu-baseline --no-extra --steps 5 > $O/r03_bench_text_L9.json 2> $O/text9.err
python bench.py --input text --history 32 --no-cpu-baseline --no-extra > $O/r03_bench_text_L1_history.json 2> $O/texth.err
python bench.py --input zipf --level 5 --no-cpu-baseline --no-extra > $O/r03_bench_zipf_L5.json 2> $O/zipf5.err
python bench.py --mode decompress --level 5 --input mixed --frame-mib 1 --no-extra > $O/r03_bench_decompress_L5_1MiB_frames_1GiB.json 2> $O/d1.err
python bench.py --mode decompress --level 5 --input mixed --frame-mib 0.0625 --no-cpu-baseline --no-extra > $O/r03_bench_decompress_L5_64KiB_frames_1GiB.json 2> $O/d2.err
python bench.py --mode decompress --level 5 --input text --frame-mib 16 --size-mib 256 --unique-mib 16 --no-cpu-baseline --no-extra > $O/r03_bench_decompress_L5_16MiB_text_frames.json 2> $O/d3.err
python tools/host_path_rate.py > $O/r03_host_path_rate.txt 2>&1
ls $O
