"""Extract the gfx950 code object from a HIP fat binary (.o or .so): python tools/extract_co.py <in> <out.co>
(the image's roc-obj tools need a perl module that is not installed).  Handles plain and CCOB-compressed clang offload bundles."""
import struct, sys, subprocess, os

def bundles(blob):
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    at = 0
    while True:
        at = blob.find(magic, at)
        if at < 0:
            return
        n, = struct.unpack_from("<Q", blob, at + 24)
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p); p += 24
            triple = blob[p:p + tl].decode(); p += tl
            yield triple, blob[at + off:at + off + size]
        at += 24

def main():
    blob = open(sys.argv[1], "rb").read()
    if b"__CLANG_OFFLOAD_BUNDLE__" not in blob and b"CCOB" in blob:
        import zlib
        at = blob.find(b"CCOB")
        ver, method = struct.unpack_from("<HH", blob, at + 4)
        hdr = 24 if ver >= 2 else 20
        # version 2: magic, version, method, total size (u32), uncompressed size (u32), hash (u64)
        total, usize = struct.unpack_from("<II", blob, at + 8)
        data = blob[at + hdr:at + total]
        if method == 0:
            blob = zlib.decompress(data)
        else:
            import tempfile
            with tempfile.NamedTemporaryFile(suffix=".zst", delete=False) as f:
                f.write(data)
            blob = subprocess.run(["zstd", "-dc", f.name], capture_output=True).stdout
            os.unlink(f.name)
    best = None
    for triple, code in bundles(blob):
        if "gfx950" in triple and len(code) > 0:
            best = code
    if best is None:
        sys.exit("no gfx950 code object found")
    open(sys.argv[2], "wb").write(best)
    print("wrote", sys.argv[2], len(best), "bytes")

main()
