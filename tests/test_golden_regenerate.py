"""Closes the golden loop: tests/golden/make_golden.py --check imports the REFERENCE (/root/reference, with the stand-ins of
tests/golden/_refstubs for its absent third-party modules), regenerates the fast subset of the fixtures into a scratch
directory and compares every array with the committed files.  The reference never travels to the GPU box: skipped there."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, REPO

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "path_planning")), reason="the reference is only present in the build container")
def test_committed_goldens_are_what_the_reference_produces():
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), "--check"], cwd=REPO, capture_output=True, text=True,
                       timeout=900)
    last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and last.startswith("golden check:") and " 0 differences" in last, (last, r.stderr[-1500:])
    n = int(last.split()[2])
    assert n >= 25   # g7 g5 g4 g1 g12 g14 g15 + eight g3 cases (the three edge cases of commit e55714b among them)


def test_every_g3_golden_has_a_spec_in_the_generator():
    """a committed g3_*.npz that make_golden.py cannot regenerate is an orphan (three were, until round 5)"""
    src = open(os.path.join(GOLDEN, "make_golden.py")).read()
    names = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith("g3_") and f.endswith(".npz"))
    assert names and all('("%s"' % n in src for n in names), [n for n in names if '("%s"' % n not in src]
