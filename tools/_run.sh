cd ${GRAFT_REPO_ROOT:-.}
bash tools/prof_tile.sh r05i > gpurun_out/prof_tile_r05i.log 2>&1
python bench.py 2> gpurun_out/bench_default.err | tail -1 > gpurun_out/bench_default.json
