"""Per-kernel resources of the built libsge_amd.so (registers, LDS, scratch, spills), read from the gfx950 code objects' metadata
notes — no GPU needed. Usage: kernel_resources.py [path/to/libsge_amd.so]. tests/test_abi.py imports `kernel_table` for its
resource guards (a kernel that starts to use scratch, or whose static LDS no longer fits the residency DESIGN.md counts on, is
caught at build time, not by a fault on the GPU box)."""
import os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_table(lib):
    """{demangled-ish kernel name: {vgpr, sgpr, lds, scratch, sgpr_spill, vgpr_spill, dynamic_stack}} of every gfx950 kernel in `lib`."""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        so = os.path.join(td, "lib.so")
        shutil.copy(lib, so)
        # llvm-objdump --offloading writes one file per bundle entry next to its input
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=td, check=True, capture_output=True)
        for f in sorted(os.listdir(td)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(td, f)], check=True, capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count:")[1:]:
                def field(k, cast=int, default=0):
                    m = re.search(r"^\s*\.%s:\s*(\S+)" % k, block, re.M)
                    return cast(m.group(1)) if m else default
                name = field("name", str, "")
                if not name:
                    continue
                out[name] = dict(vgpr=field("vgpr_count"), sgpr=field("sgpr_count"), lds=field("group_segment_fixed_size"),
                                 scratch=field("private_segment_fixed_size"), sgpr_spill=field("sgpr_spill_count"),
                                 vgpr_spill=field("vgpr_spill_count"), dynamic_stack=field("uses_dynamic_stack", str, "false") == "true",
                                 max_wg=field("max_flat_workgroup_size"))
    return out


def waves_per_simd(vgpr):
    alloc = (vgpr + 7) // 8 * 8
    return min(8, 512 // max(alloc, 8))


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "swift-game-engine_amd", "libsge_amd.so")
    tab = kernel_table(lib)
    print("%-72s %5s %5s %7s %7s %6s %6s %5s" % ("kernel", "vgpr", "sgpr", "lds B", "scratch", "sspill", "vspill", "w/SIMD"))
    for name in sorted(tab):
        r = tab[name]
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(.*", "", short)
        print("%-72s %5d %5d %7d %7d %6d %6d %5d" % (short[:72], r["vgpr"], r["sgpr"], r["lds"], r["scratch"], r["sgpr_spill"], r["vgpr_spill"], waves_per_simd(r["vgpr"])))
