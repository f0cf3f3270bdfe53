"""The CLIs keep the reference's argv/stdout surface (1d/main.cu:43-94, 2d/main.cu:97-215, 3d/main.cu:71-110)."""
import os
import subprocess

import pytest
from conftest import ROOT, has_gpu

BIN = os.path.join(ROOT, "lorastencil_amd", "bin")


def run(dim, *args):
    p = subprocess.run([os.path.join(BIN, f"lorastencil_{dim}d"), *args], capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout, p.stderr


HELP2 = ("Program name: lorastencil_2d\n"
         "Usage: lorastencil_2d shape input_size_of_first_dimension input_size_of_second_dimension time_size\n"
         "Shape: box2d1r or star2d1r or box2d3r or star2d3r\n\n")


def test_help_on_too_few_arguments_and_bad_shape(engine_built):
    rc, out, _ = run(2, "star2d1r", "64", "128")
    assert rc == 1 and out == HELP2
    rc, out, _ = run(2, "star3d1r", "64", "128", "1")  # a 3D shape is not a 2D shape
    assert rc == 1 and out == HELP2
    rc, out, _ = run(1)
    assert rc == 1 and out.startswith("Program name: lorastencil_1d\nUsage: lorastencil_1d shape input_size time_size\n"
                                      "Shape: 1d1r or 1d2r\n")
    rc, out, _ = run(3, "box3d1r", "8", "8", "8")
    assert rc == 1 and "Shape: box3d1r or star3d1r" in out


def test_integer_parsing_messages(engine_built):
    rc, _, err = run(2, "star2d1r", "abc", "128", "1")
    assert rc == 1 and err == "Invalid argument: cannot convert the parameter(s) to integer.\n"
    rc, _, err = run(1, "1d1r", "99999999999999", "1")
    assert rc == 1 and err == "Argument out of range: the parameter(s) is(are) too large.\n"


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_info_line_then_loud_failure_without_gpu(engine_built):
    rc, out, _ = run(2, "star2d1r", "64", "128", "2")
    assert out.startswith("INFO: shape = star_2d1r, m = 64, n = 128, times = 2\n")
    assert rc == 1 and "no HIP device" in out
