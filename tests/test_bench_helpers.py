"""CPU: the parts of bench.py that need no GPU -- the sysfs reader behind the line's `device` block."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_device_sample_reads_the_levels_in_force(tmp_path):
    import bench
    assert bench.device_sample(None) is None
    d = tmp_path / "dev"
    hw = d / "hwmon" / "hwmon3"
    hw.mkdir(parents=True)
    (d / "pp_dpm_sclk").write_text("0: 132Mhz\n1: 2389Mhz *\n2: 2400Mhz\n")
    (d / "pp_dpm_mclk").write_text("0: 900Mhz\n1: 2000Mhz *\n")
    (hw / "power1_average").write_text("1255000000\n")
    (hw / "power1_cap").write_text("1400000000\n")
    (hw / "temp1_input").write_text("61000\n")
    s = bench.device_sample(str(d))
    assert s == {"sclk_mhz": 2389, "mclk_mhz": 2000, "power_w": 1255.0, "power_cap_w": 1400.0, "temp_c": 61.0}


def test_device_sysfs_without_a_gpu_is_none_or_a_directory():
    import bench
    p = bench.device_sysfs(0)
    assert p is None or os.path.isdir(p)
