"""Host-only checks of the committed measurement artefacts: the bench line's `traffic` fields are read from
profiles/traffic_r*.json, so the file must hold an entry for every workload and configuration the bench reports."""
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_traffic_file_covers_every_bench_record():
    import bench
    import bench_configs
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_r*.json")))
    assert files, "no committed PMC traffic summary"
    newest = json.load(open(files[-1]))
    for tag, kernel in (("default", "vx_k_yolo_fused"), ("dense", "vx_k_yolo_fused"), ("unfused", "vx_k_yolo_decode_rows")):
        value, src = bench.load_traffic(tag)
        assert value and src == os.path.relpath(files[-1], ROOT)
        k = newest[tag]["kernels"][kernel]
        assert value == k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"]
    for tag, kernel in (("fpn_ssd_640_b64", "vx_k_prior_decode"), ("retinaface_640_b128", "vx_k_prior_decode_reg"),
                        ("yolo416_b256_dense", "vx_k_yolo_fused"), ("yolo416_b256_resize720p", "vx_k_resize_fast<true>")):
        assert tag in bench_configs.CONFIGS
        value, src = bench_configs._traffic(tag, kernel)
        assert value and value > 1e6, (tag, kernel)


def test_headline_traffic_is_close_to_the_algorithmic_bytes():
    """PMC bytes of the dominant kernel (FETCH_SIZE doubled as the guide prescribes) against the algorithmic bytes of
    one 256-image launch: a ratio far from 1 would mean wasted re-reads or a wrong correction."""
    import bench
    value, _ = bench.load_traffic("default")
    algo = 256 * bench.ALGO_BYTES_DECODE
    assert 0.95 < value / algo < 1.15, value / algo


def test_every_profile_has_its_bench_line_and_kernel_stats():
    tags = {os.path.basename(f)[:-len("_kernel_stats.csv")] for f in glob.glob(os.path.join(ROOT, "profiles", "r0[23]_*_kernel_stats.csv"))}
    assert {"r02_default", "r02_dense", "r02_unfused", "r02_retinaface", "r02_fpn_ssd", "r02_stress", "r02_resize720p"} <= tags
    assert {"r03_default", "r03_dense", "r03_unfused", "r03_retinaface", "r03_fpn_ssd", "r03_stress", "r03_stress_all",
            "r03_resize720p", "r03_dense_cfg", "r03_b1"} <= tags
    for t in tags:
        assert os.path.exists(os.path.join(ROOT, "profiles", t + "_bench.json")), t
