"""The closed form of the integration kernel's z-prefix replay (tools/micro/replay_closed_form.cpp) against the
sequential additions, bit for bit: random accumulators and increments over sixty binades, rounding ties, binade edges,
increments around half a unit in the last place, and the column geometry of the integration kernel up to 4096 planes.

The construction is exact; the phase counts it prints are why the kernels keep the sequential replay (a wavefront's
slowest accumulator needs about 25 jumps of some thirty instructions each, for six accumulators, against 3 packed
additions per replayed plane)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_closed_form_replay_equals_the_sequential_loop(tmp_path):
    exe = str(tmp_path / "replay_closed_form")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tools", "micro", "replay_closed_form.cpp")], check=True)
    out = subprocess.run([exe, "30000", "250"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL EQUAL" in out.stdout
    assert out.stdout.count(" 0 mismatches") == 5, out.stdout
    # the figure the design decision rests on: phases of a wavefront's slowest accumulator
    phases = {int(m.group(1)): float(m.group(2))
              for m in re.finditer(r"(\d+) planes\s+:.*?needs ([\d.]+) phases on average", out.stdout)}
    assert set(phases) == {512, 1024, 2048, 4096}
    assert all(15.0 < v < 40.0 for v in phases.values()), phases
