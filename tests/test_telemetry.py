"""tools/telemetry.py: the clock-state summary beside every timing and the comparability rule of the A/B scripts (no GPU needed:
the sampler is fed by hand)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import telemetry  # noqa: E402


def _sampler(rows):
    s = telemetry.Sampler.__new__(telemetry.Sampler)
    s.samples, s.source = list(rows), "test"
    return s


def test_summary_selects_the_window_and_reports_min_median_max():
    s = _sampler([(0.0, 2400.0, 1290.0, 50.0), (1.0, 2390.0, 1280.0, 51.0), (2.0, 1700.0, 300.0, 45.0), (3.0, None, None, None)])
    a = s.summary(0.0, 1.5)
    assert a["n"] == 2 and a["sclk_mhz"] == {"min": 2390.0, "median": 2395.0, "max": 2400.0}
    assert s.summary(2.5, 4.0)["sclk_mhz"] is None            # a window with no readable clock sample
    assert s.summary()["n"] == 4


def test_comparable_wants_the_same_clock_and_the_same_power():
    sus = {"sclk_mhz": {"median": 2391.0}, "power_w": {"median": 1286.0}}
    assert telemetry.comparable(sus, {"sclk_mhz": {"median": 2406.0}, "power_w": {"median": 1250.0}})
    assert not telemetry.comparable(sus, {"sclk_mhz": {"median": 1800.0}, "power_w": {"median": 1286.0}})        # clocks 25 % apart
    # the measured case (profiles/r04_clock_states.txt): 100 ms after the GPU went idle the reported clock is back, the power is not
    assert not telemetry.comparable(sus, {"sclk_mhz": {"median": 2402.0}, "power_w": {"median": 675.0}})
    assert telemetry.comparable({"sclk_mhz": {"median": 2391.0}}, {"sclk_mhz": {"median": 2400.0}})             # no power samples: clock only
    assert not telemetry.comparable(sus, {"sclk_mhz": None, "power_w": None})                                   # nothing sampled: refuse
    assert not telemetry.comparable({}, sus)


def test_sampler_thread_starts_and_stops_without_a_gpu():
    s = telemetry.Sampler(0, period_s=0.001)
    if s.hw is None:            # no amdgpu hwmon node here: the fallback would spawn rocm-smi per sample; only the bookkeeping is checked
        assert s.source == "rocm-smi"
        return
    with s:
        time.sleep(0.02)
    assert s.summary()["n"] >= 1
