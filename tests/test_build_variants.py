"""Every developer macro the kernel sources still carry compiles (a variant nothing compiles rots silently).

The timing-only ablations kept in the tree (results WRONG by construction; profiles/ab.sh builds them with PESTO_EXTRA_CXXFLAGS):
  PESTO_ABL_NOSPLIT   no residual in the f16 hi/lo split              PESTO_ABL_NOELU     one v_max instead of exp + fma + med3
  PESTO_ABL_NOGATHER  every neighbour gather from 8 hot rows            PESTO_ABL_NOCENLD   centre records / own state from 16 hot records
  PESTO_ABL_NOPREPST  the prepare phase's record stores alias           PESTO_ABL_NONODE    the node waves only keep their queues moving
  PESTO_ABL_NOPREP    no prepare phase at all (pesto_api.hip)
and one instrument (results unchanged): PESTO_DEV_TIMELINE - wall-clock stamps of every workgroup of a layer launch (profiles/dev/timeline.py)
and one A/B form (same bits): PESTO_SPLIT_MIXLO - the f16 hi/lo split's residual as v_fma_mix{lo,hi}_f16 per element, as shipped until round 5
Checked with `hipcc -fsyntax-only` for gfx950 (semantic analysis instantiates every kernel template the launchers use): seconds per
macro, no GPU. Also: no OTHER `PESTO_*` preprocessor switch is left in the layer-kernel sources."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "pesto_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
LAYER_SOURCES = ["pesto_edge.hip", "pesto_node.hip", "pesto_mfma_common.h", "pesto_fin_rendezvous.inc", "pesto_edge_node_waves.inc"]
ABLATIONS = {"PESTO_ABL_NOSPLIT": "pesto_edge.hip", "PESTO_ABL_NOELU": "pesto_edge.hip", "PESTO_ABL_NOGATHER": "pesto_edge.hip",
             "PESTO_ABL_NOCENLD": "pesto_edge.hip", "PESTO_ABL_NOPREPST": "pesto_edge.hip", "PESTO_ABL_NONODE": "pesto_edge.hip",
             "PESTO_ABL_NOPREP": "pesto_api.hip", "PESTO_DEV_TIMELINE": "pesto_edge.hip", "PESTO_SPLIT_MIXLO": "pesto_edge.hip"}


def _syntax(source, *defines):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=on", "-x", "hip", "--cuda-device-only",
           "-fsyntax-only"] + [f"-D{d}" for d in defines] + [os.path.join(CSRC, source)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("macro", sorted(ABLATIONS))
def test_every_kept_ablation_compiles(macro):
    p = _syntax(ABLATIONS[macro], macro)
    assert p.returncode == 0, p.stderr[-1500:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_all_edge_ablations_together_and_the_node_file():
    p = _syntax("pesto_edge.hip", *[m for m, s in ABLATIONS.items() if s == "pesto_edge.hip" and m.startswith("PESTO_ABL_")])
    assert p.returncode == 0, p.stderr[-1500:]
    p = _syntax("pesto_node.hip", "PESTO_ABL_NOSPLIT", "PESTO_ABL_NOELU")
    assert p.returncode == 0, p.stderr[-1500:]


def test_no_other_developer_switch_is_left_in_the_layer_kernels():
    """the measured-and-dropped variants of rounds 3 - 5 were deleted in round 6 (git history and profiles/HISTORY.md keep them): what the
    preprocessor can still switch in the layer-kernel sources is exactly the ablation list above"""
    found = set()
    for f in LAYER_SOURCES:
        for m in re.finditer(r"^\s*#\s*(?:if|ifdef|ifndef|elif)\b[^\n]*?\b(PESTO_[A-Z0-9_]+)", open(os.path.join(CSRC, f)).read(), re.M):
            found.add(m.group(1))
    assert found <= set(ABLATIONS), sorted(found - set(ABLATIONS))
    lines = sum(len(open(os.path.join(CSRC, f)).read().split("\n")) for f in LAYER_SOURCES)
    assert max(len(open(os.path.join(CSRC, f)).read().split("\n")) for f in LAYER_SOURCES) < 2500 and lines < 3000, lines
