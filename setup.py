"""Build hook: `pip install .` / `python setup.py build_py` compiles the HIP library for gfx950 first
(`make -C paroquant_amd/csrc`, hipcc cross-compiles without a GPU) so that the wheel ships
paroquant_amd/_lib/libparo_mi355x.so.  Metadata lives in pyproject.toml."""
import os
import subprocess

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


class build_py_with_hip(build_py):
    def run(self):
        jobs = str(max(1, min(8, os.cpu_count() or 1)))
        subprocess.run(["make", "-j", jobs, "-C", os.path.join(ROOT, "paroquant_amd", "csrc")], check=True)
        super().run()


setup(cmdclass={"build_py": build_py_with_hip})
