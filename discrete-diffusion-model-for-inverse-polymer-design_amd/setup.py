"""Editable install of the `bidiff` package (`pip install -e .` from this directory).

The HIP library is not built here: run `make -C csrc` first; `bidiff/_lib.py` loads `lib/libbidiff_hip.so`
next to the package, so the install has to be editable (the sources stay where they are)."""
from setuptools import find_packages, setup

setup(
    name="bidiff-mi355x",
    version="0.2.0",
    description="Bi_Diffusion reverse sampling and training step on MI355X (gfx950): drop-in classes over a C-ABI HIP library",
    packages=find_packages(include=["bidiff", "bidiff.*"]),
    python_requires=">=3.10",
    install_requires=["torch", "numpy"],
)
