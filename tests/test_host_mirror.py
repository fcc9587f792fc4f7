"""The C++ host mirror (swift-game-engine_amd/host/sge_host.hpp): it must compile against the public header and link
against the product library (CPU check), and behave on a GPU (tests/cpp/host_mirror_smoke.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swift-game-engine_amd")
SRC = os.path.join(ROOT, "tests", "cpp", "host_mirror_smoke.cpp")


def build(tmp_path, sge):
    exe = str(tmp_path / "host_mirror_smoke")
    cmd = ["g++", "-std=c++17", "-Wall", "-Wextra", SRC, "-I" + os.path.join(ROOT, "include"), "-L" + PKG, "-lsge_amd",
           "-Wl,-rpath," + PKG, "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_host_mirror_compiles_and_links(tmp_path, sge):
    assert os.path.exists(build(tmp_path, sge))


@pytest.mark.gpu
def test_host_mirror_on_gpu(tmp_path, sge):
    out = subprocess.run([build(tmp_path, sge)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host mirror smoke ok" in out.stdout


ALLGATHER_SRC = os.path.join(ROOT, "tests", "cpp", "allgather_smoke.cpp")


def build_allgather(tmp_path):
    exe = str(tmp_path / "allgather_smoke")
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    subprocess.check_call([hipcc, "-std=c++17", ALLGATHER_SRC, "-I" + os.path.join(ROOT, "include"), "-L" + PKG, "-lsge_amd", "-lrccl",
                           "-Wl,-rpath," + PKG, "-o", exe])
    return exe


def test_allgather_host_sample_compiles_and_links(tmp_path, sge):
    """The C host sequence of the character-vs-character exchange (sge_agents_allgather with the caller's ncclComm_t)."""
    assert os.path.exists(build_allgather(tmp_path))


@pytest.mark.gpu
def test_allgather_host_sample_on_gpu(tmp_path, sge):
    out = subprocess.run([build_allgather(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "allgather smoke ok" in out.stdout


def test_swift_binding_sources_refer_only_to_declared_c_names():
    """No Swift toolchain here: what can be checked of swift-game-engine_amd/host/swift/*.swift is that every C function, struct and
    constant they name exists in include/sge_amd.h and that the files are balanced."""
    import glob
    import re
    hdr = open(os.path.join(ROOT, "include", "sge_amd.h")).read()
    declared = set(re.findall(r"\b(sge_[a-z_0-9]+)\b", hdr)) | set(re.findall(r"\b(SGE_[A-Z_0-9]+)\b", hdr))
    files = sorted(glob.glob(os.path.join(ROOT, "swift-game-engine_amd", "host", "swift", "*.swift")))
    assert len(files) >= 5
    for f in files:
        text = re.sub(r"//.*", "", open(f).read())
        text = re.sub(r'"(\\.|[^"\\])*"', '""', text)
        for a, b in ("{}", "()", "[]"):
            assert text.count(a) == text.count(b), (os.path.basename(f), a)
        used = set(re.findall(r"\b(sge_[a-z_0-9]+)\b", text)) | set(re.findall(r"\b(SGE_[A-Z_0-9]+)\b", text))
        assert not (used - declared), (os.path.basename(f), sorted(used - declared))

