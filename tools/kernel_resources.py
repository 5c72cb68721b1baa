"""Registers / scratch / LDS of every gfx950 kernel inside a host object or shared library (no `hipcc -S` rebuild needed).
   python tools/kernel_resources.py savitzky-golay-filter_amd/build/sg_2d_roll_g1.o [substring ...]
Finds the clang offload bundles in the file, writes each gfx950 code object to /tmp and reads its kernel metadata notes."""
import re, struct, subprocess, sys, tempfile, os

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def unpacked(blob):
    """The file with every zstd-compressed bundle (`--offload-compress`: magic CCOB, version 2 / 3 header, then the zstd frame) replaced by the
    plain __CLANG_OFFLOAD_BUNDLE__ it holds -- appended, the scan below does not care where a bundle sits."""
    if blob.find(b"CCOB") < 0:
        return blob
    import ctypes
    z = ctypes.CDLL("libzstd.so.1")
    z.ZSTD_decompress.restype = ctypes.c_size_t
    z.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    out, pos = [blob], 0
    while True:
        pos = blob.find(b"CCOB", pos)
        if pos < 0:
            return b"".join(out)
        version, method = struct.unpack_from("<HH", blob, pos + 4)
        if version >= 3:
            total, raw = struct.unpack_from("<QQ", blob, pos + 8)
            head = 32
        else:
            total, raw = struct.unpack_from("<II", blob, pos + 8)
            head = 24
        if method == 1 and head < total <= len(blob) - pos and raw < (1 << 31):
            dst = ctypes.create_string_buffer(raw)
            got = z.ZSTD_decompress(dst, raw, blob[pos + head:pos + total], total - head)
            if got == raw:
                out.append(dst.raw)
        pos += 4


def code_objects(blob):
    blob = unpacked(blob)
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", blob, pos + 24)
        off = pos + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            if "gfx950" in triple and size:
                yield blob[pos + o:pos + o + size]
        pos += 24


def main():
    path, filters = sys.argv[1], sys.argv[2:]
    blob = open(path, "rb").read()
    for co in code_objects(blob):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(co)
        txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        os.unlink(f.name)
        for block in txt.split("- .agpr_count:")[1:]:
            def g(key):
                m = re.search(r"\." + key + r":\s+(\S+)", block)
                return m.group(1) if m else "?"
            name = g("name")
            if filters and not all(s in name for s in filters):
                continue
            demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            demangled = re.sub(r"\(.*", "", demangled)
            print(f"vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} scratch {g('private_segment_fixed_size'):>4} spill {g('vgpr_spill_count'):>3} lds {g('group_segment_fixed_size'):>6}  {demangled}")


if __name__ == "__main__":
    main()
