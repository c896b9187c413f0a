"""What bench.py quotes from profiles/ is derived from the measurements committed there, not typed in: the C4 floors
(tools/c4_floors.py -> profiles/r04_c4_floors.json -> bench.c4_issue), and the HBM traffic of the headline kernel."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_c4_floors_follow_from_the_committed_measurements():
    tool = _load(os.path.join(ROOT, "tools", "c4_floors.py"), "c4_floors")
    committed = json.load(open(os.path.join(ROOT, "profiles", "r04_c4_floors.json")))
    wide, pipe = tool.wide_floor(), tool.pipeline_floor()
    assert abs(wide["floor_cycles_per_sample"] - committed["wide"]["floor_cycles_per_sample"]) < 1e-9
    assert abs(pipe["floor_cycles_per_sample"] - committed["pipeline"]["floor_cycles_per_sample"]) < 1e-9
    sq = json.load(open(os.path.join(ROOT, "profiles", "r04_c4_wide_sq_counters.json")))
    want = (sq["valu_wave_insts_per_voice_sample"] + sq["lds_wave_insts_per_voice_sample"]) * tool.ISSUE_CYCLES
    assert abs(wide["floor_cycles_per_sample"] - want) < 1e-9
    # ... and the counters say what they are said to say: instructions per voice-sample from the raw sums
    per = sq["SQ_INSTS_VALU"] / (sq["voices"] / 64.0 * sq["block_size"] * sq["blocks_per_launch"])
    assert abs(per - sq["valu_wave_insts_per_voice_sample"]) < 1e-6


def test_bench_quotes_the_committed_floor():
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_module")
    committed = json.load(open(os.path.join(ROOT, "profiles", "r04_c4_floors.json")))
    wide = bench.c4_issue(65536, 512, 34.0)
    assert abs(wide["floor_us_per_block"] - committed["wide"]["floor_us_per_block_of_512"]) < 1e-9
    assert abs(wide["frac_of_floor"] - wide["floor_us_per_block"] / 34.0) < 1e-12
    pipe = bench.c4_issue(8192, 512, 19.9)
    assert abs(pipe["floor_us_per_block"] - committed["pipeline"]["floor_us_per_block_of_512"]) < 1e-9
    twice = bench.c4_issue(131072, 512, 60.0)  # two wavefronts per SIMD share its issue
    assert abs(twice["floor_us_per_block"] - 2 * wide["floor_us_per_block"]) < 1e-9


def test_headline_traffic_comes_from_this_rounds_counters():
    bench = _load(os.path.join(ROOT, "bench.py"), "bench_module2")
    got = bench.traffic_from_profiles(16384, 512, "f32")
    assert got is not None
    prof = json.load(open(os.path.join(ROOT, "profiles", "r04_hbm_traffic.json")))
    k = prof["voice_pipe_kernel"]
    assert abs((k["FETCH_SIZE_KB_per_launch"] + k["WRITE_SIZE_KB_per_launch"]) * 1024.0 - k["hbm_bytes_per_launch"]) < 1.0
