set -e
mkdir -p gpurun_out/r03_b
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "fresnel or matches_oracle or random_scenes or xyz_film" > gpurun_out/r03_b/pytest.log 2>&1 || true
tail -4 gpurun_out/r03_b/pytest.log
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-oneshot > gpurun_out/r03_b/bench_pair.json 2> gpurun_out/r03_b/bench_pair.err
DRT_NO_PAIR_ROWS=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-oneshot > gpurun_out/r03_b/bench_nopair.json 2> gpurun_out/r03_b/bench_nopair.err
ONLY=4 timeout -k 10 300 python tools/run_configs.py > gpurun_out/r03_b/cfg4_pair.json 2>&1
DRT_NO_PAIR_ROWS=1 ONLY=4 timeout -k 10 300 python tools/run_configs.py > gpurun_out/r03_b/cfg4_nopair.json 2>&1
python - <<'PY'
import json
for n in ("pair","nopair"):
    j=json.load(open("gpurun_out/r03_b/bench_%s.json"%n)); print(n, j["value"], j["roofline"]["kernel_ms_per_step"])
    print(open("gpurun_out/r03_b/cfg4_%s.json"%n).read().strip())
PY
