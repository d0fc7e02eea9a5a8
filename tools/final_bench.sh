set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python bench.py > $O/r02_j_bench_zipf_L1.json 2> $O/zipf.err
python bench.py --input text > $O/r02_j_bench_text_L1.json 2> $O/text.err
python bench.py --input text --history 32 --no-cpu-baseline > $O/r02_j_bench_text_L1_history.json 2> $O/texth.err
python bench.py --input mixed --level 5 > $O/r02_j_bench_mixed_L5.json 2> $O/mixed.err
python bench.py --input text --level 3 --no-cpu-baseline > $O/r02_j_bench_text_L3.json 2> $O/text3.err
python bench.py --input text --level 9 --no-cpu-baseline --steps 5 > $O/r02_j_bench_text_L9.json 2> $O/text9.err
python bench.py --input text --level 5 --no-cpu-baseline > $O/r02_j_bench_text_L5.json 2> $O/text5.err
python bench.py --mode decompress --level 5 --input mixed --frame-mib 1 > $O/r02_j_bench_decompress_L5_1MiB_frames_1GiB.json 2> $O/d1.err
python bench.py --mode decompress --level 5 --input mixed --frame-mib 0.0625 --no-cpu-baseline > $O/r02_j_bench_decompress_L5_64KiB_frames_1GiB.json 2> $O/d2.err
python bench.py --mode decompress --level 5 --input mixed --frame-mib 1 --size-mib 4096 --no-cpu-baseline > $O/r02_j_bench_decompress_L5_1MiB_frames_4GiB.json 2> $O/d3.err
python bench.py --input text --size-mib 10 --steps 50 --no-cpu-baseline > $O/r02_j_bench_text_L1_10MiB.json 2> $O/t10.err
ls $O
