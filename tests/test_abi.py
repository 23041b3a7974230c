"""CPU-only checks of the drop-in boundary: the shared library loads and exports every
symbol that include/mtlssl_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    from mtl_ssl_amd import lib
    return lib


def test_library_exports_every_declared_symbol(built):
    protos = built.parse_header()
    assert len(protos) >= 30
    cdll = ctypes.CDLL(built.LIB_PATH)
    for name in protos:
        assert hasattr(cdll, name), name


def test_abi_version_and_error_string(built):
    L = built.lib()
    import re
    want = int(re.search(r"#define\s+MTLSSL_ABI_VERSION\s+(\d+)", open(built.HEADER).read()).group(1))
    assert L.abi_version() == want >= 6
    assert isinstance(L.last_error(), bytes)


def test_header_cites_reference_for_every_family():
    text = open(os.path.join(ROOT, "include", "mtlssl_hip.h")).read()
    for needle in ("resnet_utils.py", "grid_anchor_generator.py", "faster_rcnn_box_coder.py",
                   "argmax_matcher.py", "target_assigner.py", "balanced_positive_negative_sampler.py",
                   "faster_rcnn_meta_arch.py", "losses.py", "learning.py"):
        assert needle in text, needle


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
    cpu_baseline leg may import it."""
    import re
    pkg = os.path.join(ROOT, "mtl_ssl_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


def test_committed_pmc_traffic_was_taken_from_the_current_roofline_kernel():
    """bench.py's roofline.traffic is the committed result of separate PMC passes (counters cannot be read inside
    the timed process): the file carries a digest of the kernel sources it was measured on, and a change to
    conv_mfma.h without re-running tools/pmc_bench.sh + tools/pmc_summary.py fails here."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from pmc_summary import kernel_source_sha16
    sys.path.insert(0, root)
    import bench
    path = bench._latest_profile("pmc_traffic.json")          # the file bench.py's roofline.traffic is read from
    assert path is not None, "no profiles/rNN_pmc_traffic.json committed"
    d = json.load(open(path))
    assert d.get("kernel_source_sha16") == kernel_source_sha16(), \
        "%s is stale: regenerate it with tools/config_evidence.sh <tag> pmc (or tools/pmc_bench.sh + tools/pmc_summary.py)" \
        % os.path.relpath(path, root)
