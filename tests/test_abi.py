"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
the public headers declare, keeps the reference's macros, and fails loudly without a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

import lzs_compression_amd as lzs

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
INC = os.path.join(ROOT, "include")


def _declared_functions():
    names = []
    for hdr in ("lzs/lzs.h", "lzs/lzs_batch.h"):
        text = open(os.path.join(INC, hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"static inline[^{]*\{[^}]*\}", "", text)       # header-only, as in the reference (lzs.h:239)
        names += re.findall(r"\b(lzs_[a-z_]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported():
    lib = lzs.lib()
    declared = _declared_functions()
    assert {"lzs_compress", "lzs_decompress", "lzs_compress_batch_device",
            "lzs_decompress_batch_device", "lzs_compact_device", "lzs_compress_batch",
            "lzs_decompress_batch", "lzs_backend_info", "lzs_last_error"} <= set(declared)
    for name in declared:
        assert hasattr(lib, name), name


def test_headers_compile_as_c99_and_keep_reference_macros(tmp_path):
    """A C99 caller of the reference's one-shot interface compiles unchanged against our
    header; LZS_COMPRESSED_MAX/LZS_DECOMPRESSED_MAX keep their values (lzs.h:77,81)."""
    src = tmp_path / "t.c"
    src.write_text(r'''
#include "lzs.h"
#include "lzs_batch.h"
#include <stdio.h>
int main(void) {
    size_t (*c)(uint8_t *, size_t, const uint8_t *, size_t) = lzs_compress;
    size_t (*d)(uint8_t *, size_t, const uint8_t *, size_t) = lzs_decompress;
    printf("%u %u %u %u %u %d\n", (unsigned)LZS_COMPRESSED_MAX(65536u), (unsigned)LZS_COMPRESSED_MAX(4096u),
           (unsigned)LZS_DECOMPRESSED_MAX(10u), (unsigned)LZS_MAX_HISTORY_SIZE,
           (unsigned)LZS_MAX_LOOK_AHEAD_LEN, (c != 0) && (d != 0));
    return 0;
}
''')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", f"-I{INC}/lzs", str(src),
                    f"-L{ROOT}/lzs_compression_amd", "-llzs",
                    f"-Wl,-rpath,{ROOT}/lzs_compression_amd", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert out == ["73731", "4611", "160", "2047", "15", "1"]


def test_library_carries_the_reference_soname():
    """libtool -version-info 4:0:0 (reference c/configure.ac:17, c/src/liblzs/Makefile.am:15) gives
    liblzs.so.4: a program linked with -llzs against either build asks for that name at run time."""
    so = os.path.join(ROOT, "lzs_compression_amd", "liblzs.so")
    dyn = subprocess.run(["readelf", "-d", so], capture_output=True, text=True, check=True).stdout
    assert "Library soname: [liblzs.so.4]" in dyn
    assert os.path.exists(so + ".4")


def test_headers_are_usable_from_cxx(tmp_path):
    """Both headers in a C++ translation unit (extern "C" guards, no C-only constructs), linked
    against the library."""
    src = tmp_path / "t.cc"
    src.write_text(r'''
#include "lzs.h"
#include "lzs_batch.h"
#include <cstdio>
int main() {
    LzsCompressParameters_t c; LzsDecompressParameters_t d;
    lzs_compress_init(&c); lzs_decompress_init(&d);
    static_assert(sizeof(c) == 14432 && sizeof(d) == 2096, "the reference's sizes");
    std::printf("%d %d\n", (int)c.status, (int)d.status);
    return lzs_last_error() == nullptr;
}
''')
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", f"-I{INC}/lzs", str(src),
                    f"-L{ROOT}/lzs_compression_amd", "-llzs",
                    f"-Wl,-rpath,{ROOT}/lzs_compression_amd", "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split() == ["0", "0"]


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lzs.LzsError) as e:
        lzs.compress(b"hello hello hello")
    assert "no HIP device" in str(e.value) and "no CPU codec" in str(e.value)
    with pytest.raises(lzs.LzsError):
        lzs.backend_info()
    # the raw C call returns 0 bytes and leaves the buffer alone
    lib = lzs.lib()
    dst = ctypes.create_string_buffer(b"\xAA" * 32, 32)
    src = ctypes.create_string_buffer(b"abcabcabc", 9)
    assert lib.lzs_compress(ctypes.addressof(dst), 32, ctypes.addressof(src), 9) == 0
    assert dst.raw == b"\xAA" * 32
    assert lib.lzs_last_error()


def test_incremental_parameter_blocks_keep_the_reference_layout(tmp_path):
    """LzsCompressParameters_t / LzsDecompressParameters_t: public members at the reference's
    offsets, whole structs of the reference's sizes (c/src/liblzs/lzs.h:101-134, 180-211:
    14432 and 2096 bytes with gcc on x86-64), the status flags with the reference's values
    (:90-98, :168-176).  Where the reference is present, measured against its own header."""
    prog = r'''
#include <stddef.h>
#include <stdio.h>
#include "lzs.h"
int main(void) {
    printf("%zu %zu %zu %zu %zu %zu ", sizeof(LzsCompressParameters_t), offsetof(LzsCompressParameters_t, inPtr),
           offsetof(LzsCompressParameters_t, outPtr), offsetof(LzsCompressParameters_t, inLength),
           offsetof(LzsCompressParameters_t, outLength), offsetof(LzsCompressParameters_t, status));
    printf("%zu %zu %zu %zu %zu %zu ", sizeof(LzsDecompressParameters_t), offsetof(LzsDecompressParameters_t, inPtr),
           offsetof(LzsDecompressParameters_t, outPtr), offsetof(LzsDecompressParameters_t, inLength),
           offsetof(LzsDecompressParameters_t, outLength), offsetof(LzsDecompressParameters_t, status));
    printf("%zu %zu %zu ", sizeof(LzsSimpleCompressParameters_t), offsetof(LzsSimpleCompressParameters_t, outLength),
           offsetof(LzsSimpleCompressParameters_t, status));
    printf("%d %d %d %d %d %d %d %d %d %d\n", LZS_C_STATUS_INPUT_STARVED, LZS_C_STATUS_INPUT_FINISHED, LZS_C_STATUS_END_MARKER,
           LZS_C_STATUS_NO_OUTPUT_BUFFER_SPACE, LZS_C_STATUS_ERROR, LZS_D_STATUS_INPUT_STARVED, LZS_D_STATUS_INPUT_FINISHED,
           LZS_D_STATUS_END_MARKER, LZS_D_STATUS_NO_OUTPUT_BUFFER_SPACE, LZS_D_STATUS_ERROR);
    return 0;
}
'''
    src = tmp_path / "layout.c"
    src.write_text(prog)

    def measure(incdir):
        exe = tmp_path / ("layout_" + str(abs(hash(incdir))))
        subprocess.run(["gcc", "-std=c99", f"-I{incdir}", str(src), "-o", str(exe)], check=True)
        return subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()

    ours = measure(f"{INC}/lzs")
    assert ours == ["14432", "0", "8", "16", "24", "32", "2096", "0", "8", "16", "24", "32", "2112", "24", "32",
                    "1", "2", "4", "8", "16", "1", "2", "4", "8", "16"]
    ref_inc = "/root/reference/c/src/liblzs"
    if os.path.exists(os.path.join(ref_inc, "lzs.h")):
        assert measure(ref_inc) == ours
    assert ctypes.sizeof(lzs.api.CompressParameters) == 14432 and ctypes.sizeof(lzs.api.DecompressParameters) == 2096


def test_incremental_calls_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    c = lzs.IncrementalCompressor()
    with pytest.raises(lzs.LzsError) as e:
        c.step(b"hello hello hello hello hello", 100, True)
    assert "no HIP device" in str(e.value)
    with pytest.raises(lzs.LzsError):                     # also for a piece that would only be collected
        lzs.IncrementalCompressor().step(b"abc", 100, False)
    d = lzs.IncrementalDecompressor()
    with pytest.raises(lzs.LzsError):
        d.step(bytes.fromhex("30e07c3000"), 100)
    # nothing written; the input is dropped and STARVED reported next to ERROR, so that a caller who
    # never looks at ERROR (the reference's tool loops: utils/lzs-decompress.c:82-121) runs out of
    # input instead of calling for ever with the same bytes
    assert d.params.inLength == 0 and d.params.outLength == 100
    assert d.params.status == lzs.api.STATUS_ERROR | lzs.api.STATUS_INPUT_STARVED | lzs.api.STATUS_INPUT_FINISHED
    c2 = lzs.IncrementalCompressor()
    with pytest.raises(lzs.LzsError):
        c2.step(b"x" * 5000, 100, False)
    assert c2.params.inLength == 0 and c2.params.status & lzs.api.STATUS_END_MARKER and c2.params.status & lzs.api.STATUS_ERROR
    # ... except what needs no codec: an empty call on an empty queue is just "starved"
    assert lzs.IncrementalDecompressor().step(b"", 10) == (b"", 0, lzs.api.STATUS_INPUT_STARVED | lzs.api.STATUS_INPUT_FINISHED)


def test_argument_failures_of_the_incremental_calls_are_terminal_too():
    """A NULL buffer with a non-zero length (no device needed to see that): ERROR, and -- like every
    other failure -- the input is dropped and END_MARKER / STARVED set, so that the loops of the
    reference's tools, which never look at ERROR (utils/lzs-compress.c:91-134,
    utils/lzs-decompress.c:82-121), end instead of calling again with the same arguments."""
    L = lzs.lib()
    A = lzs.api
    c = A.CompressParameters()
    L.lzs_compress_init_full(ctypes.addressof(c))
    out = ctypes.create_string_buffer(64)
    c.inPtr, c.inLength, c.outPtr, c.outLength = None, 10, ctypes.addressof(out), 64
    assert L.lzs_compress_incremental(ctypes.addressof(c), False) == 0
    assert c.status & A.STATUS_ERROR and c.status & A.STATUS_END_MARKER and c.inLength == 0
    assert "NULL buffer" in lzs.last_error()
    s = A.SimpleCompressParameters()
    L.lzs_simple_compress_init(ctypes.addressof(s))
    s.inPtr, s.inLength, s.outPtr, s.outLength = None, 3, ctypes.addressof(out), 64
    assert L.lzs_simple_compress_incremental(ctypes.addressof(s), True) == 0
    assert s.status & A.STATUS_ERROR and s.status & A.STATUS_END_MARKER and s.inLength == 0
    d = A.DecompressParameters()
    L.lzs_decompress_init(ctypes.addressof(d))
    src = ctypes.create_string_buffer(b"\x30\xe0\x00", 3)
    d.inPtr, d.inLength, d.outPtr, d.outLength = ctypes.addressof(src), 3, None, 8
    assert L.lzs_decompress_incremental(ctypes.addressof(d)) == 0
    assert d.status == A.STATUS_ERROR | A.STATUS_INPUT_STARVED | A.STATUS_INPUT_FINISHED and d.inLength == 0


def test_the_file_tools_exit_non_zero_without_a_gpu(tmp_path):
    """The bundled tools must not end with exit code 0 and an empty or cut file when the device is
    missing (silent data loss)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    src = tmp_path / "in.txt"
    src.write_bytes(b"hello hello hello hello " * 100)
    bin_dir = os.path.join(ROOT, "lzs_compression_amd", "bin")
    for tool, arg in (("lzs-compress", str(src)), ("lzs-decompress", os.path.join(ROOT, "tests", "golden", "text_4k.lzs"))):
        r = subprocess.run([os.path.join(bin_dir, tool), arg, str(tmp_path / "out.bin")], capture_output=True, text=True)
        assert r.returncode != 0, (tool, r.stdout, r.stderr)
        assert "no HIP device" in r.stderr or "failed" in r.stderr, r.stderr


def test_product_does_not_reference_the_oracle():
    """The shipped package and its native sources never import, link or open oracle/."""
    pkg = os.path.join(ROOT, "lzs_compression_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".inc", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in text and "liblzs_oracle" not in text \
                    and "oracle/" not in text.replace("SURVEY", ""), os.path.join(dirpath, f)
    so = os.path.join(pkg, "liblzs.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed


def test_every_symbol_of_the_reference_library_is_exported():
    """The library carries the reference's soname (liblzs.so.4), so a program linked against the
    reference must find every function it may call: all ten of c/src/liblzs/lzs.h:218-232, the
    lzs_simple_* trio included."""
    so = os.path.join(ROOT, "lzs_compression_amd", "liblzs.so")
    ours = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout.splitlines() if " T " in l}
    want = {"lzs_compress", "lzs_compress_init_quick", "lzs_compress_init_full", "lzs_compress_incremental",
            "lzs_simple_compress", "lzs_simple_compress_init", "lzs_simple_compress_incremental",
            "lzs_decompress", "lzs_decompress_init", "lzs_decompress_incremental"}
    assert want <= ours, want - ours
    ref = os.path.join(ROOT, "oracle", "_ref", "liblzs_ref.so")
    if os.path.exists(ref):
        theirs = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", ref], capture_output=True, text=True, check=True).stdout.splitlines() if " T " in l and "lzs_" in l}
        assert theirs <= ours, theirs - ours


def _pkg_config(pc_dir, *args):
    """`pkg-config <args> liblzs` -- the real one where it is installed, else the few lines of it this needs (variable
    substitution in a .pc file, Cflags / Libs)."""
    import shutil
    exe = shutil.which("pkg-config") or shutil.which("pkgconf")
    if exe:
        env = dict(os.environ, PKG_CONFIG_PATH=pc_dir)
        return subprocess.run([exe, *args, "liblzs"], capture_output=True, text=True, check=True, env=env).stdout.split()
    var, field = {}, {}
    for ln in open(os.path.join(pc_dir, "liblzs.pc")):
        ln = ln.strip()
        if ":" in ln and (" " not in ln.split(":", 1)[0]):
            k, v = ln.split(":", 1)
            field[k] = v.strip()
        elif "=" in ln:
            k, v = ln.split("=", 1)
            var[k] = v
    def expand(t):
        for _ in range(8):
            for k, v in var.items():
                t = t.replace("${%s}" % k, v)
        return t
    out = []
    if "--cflags" in args:
        out += expand(field["Cflags"]).split()
    if "--libs" in args:
        out += expand(field["Libs"]).split()
    return out


def test_install_layout_and_pkg_config(tmp_path):
    """`make install PREFIX=` gives the reference's install layout (c/src/liblzs/Makefile.am:11-15, liblzs.pc.in:6-10):
    <prefix>/include/lzs/lzs.h, <prefix>/lib/liblzs.so.4 + the development link, <prefix>/lib/pkgconfig/liblzs.pc, the two
    tools; and a program written against the reference -- its own test-lzs.c where /root/reference is there, else a
    caller of the same two functions -- builds with nothing but `pkg-config --cflags --libs liblzs`, finds liblzs.so.4
    at run time and passes (on the small calls' host route here, where there is no device)."""
    prefix = tmp_path / "prefix"
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "lzs_compression_amd", "csrc"), "install", f"PREFIX={prefix}"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for rel in ("include/lzs/lzs.h", "include/lzs/lzs_batch.h", "lib/liblzs.so.4", "lib/liblzs.so", "lib/pkgconfig/liblzs.pc",
                "bin/lzs-compress", "bin/lzs-decompress"):
        assert (prefix / rel).exists(), rel
    assert os.readlink(prefix / "lib" / "liblzs.so") == "liblzs.so.4"
    flags = _pkg_config(str(prefix / "lib" / "pkgconfig"), "--cflags", "--libs")
    assert f"-I{prefix}/include/lzs" in flags and f"-L{prefix}/lib" in flags and "-llzs" in flags
    ref_test = "/root/reference/c/src/test/test-lzs.c"
    if os.path.exists(ref_test):
        unity = "/root/reference/c/src/test/unity"
        srcs, extra, expect = [ref_test, os.path.join(unity, "unity.c")], [f"-I{unity}", "-w"], "2 Tests 0 Failures 0 Ignored"
    else:
        src = tmp_path / "caller.c"
        src.write_text('#include <stdio.h>\n#include <string.h>\n#include "lzs.h"\n'
                       'int main(void) { uint8_t c[LZS_COMPRESSED_MAX(600)], d[600], x[600]; memset(x, 88, 600);\n'
                       '  size_t n = lzs_compress(c, sizeof c, x, 600), m = lzs_decompress(d, 600, c, n);\n'
                       '  printf("%zu %zu %d\\n", n, m, memcmp(d, x, 600)); return !(n == (9 + 2 + 7 + 160 + 9 + 7) / 8 && m == 600); }\n')
        srcs, extra, expect = [str(src)], [], "24 600 0"
    exe = tmp_path / "user_program"
    r = subprocess.run(["gcc", "-O2", *extra, *srcs, *flags, "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    needed = subprocess.run(["readelf", "-d", str(exe)], capture_output=True, text=True).stdout
    assert "liblzs.so.4" in needed
    env = dict(os.environ, LD_LIBRARY_PATH=f"{prefix}/lib:" + os.environ.get("LD_LIBRARY_PATH", ""), LZS_ROUTE="host")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and expect in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    # the installed tools find the installed library by themselves ($ORIGIN/../lib)
    data = tmp_path / "in.bin"
    data.write_bytes(b"pkg-config " * 3000)
    r = subprocess.run([str(prefix / "bin" / "lzs-compress"), str(data), str(tmp_path / "out.lzs")], capture_output=True, text=True,
                       env={k: v for k, v in os.environ.items() if k != "LD_LIBRARY_PATH"})
    assert "liblzs.so" not in r.stderr or "cannot open shared object" not in r.stderr, r.stderr


def test_a_batch_call_refuses_one_array_for_the_lengths_read_and_written():
    """d_out_len[] is scratch of a compress launch (every block's class is left there before any workgroup reads d_in_len[]):
    handing the same array as both is an argument error, before anything is launched (include/lzs/lzs_batch.h)."""
    A = lzs.api
    L = A.lib()
    fake = ctypes.c_void_p(0x1000)
    for name in ("lzs_compress_batch_device", "lzs_decompress_batch_device"):
        rc = getattr(L, name)(fake, 128, 100, fake, fake, 128, fake, 64, 4, None)
        assert rc == A.LZS_E_ARG and "same array" in A.last_error(), (name, rc, A.last_error())


def test_the_release_call_is_exported_and_harmless_on_a_thread_that_keeps_nothing():
    """lzs_release_thread_cache() (include/lzs/lzs_batch.h; the reference keeps nothing after return, lzs.h:218,229): callable
    on a thread that never staged anything, twice, without a device."""
    import threading
    L = lzs.api.lib()
    errors = []

    def worker():
        try:
            L.lzs_release_thread_cache()
            L.lzs_release_thread_cache()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))
    t = threading.Thread(target=worker)
    t.start()
    t.join()
    L.lzs_release_thread_cache()
    assert not errors
