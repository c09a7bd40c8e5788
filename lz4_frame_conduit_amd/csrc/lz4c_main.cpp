// mi355x-lz4c -- command-line driver with the shape of the reference's `lz4-frame-conduit-exe`
// (/root/reference/app/Main.hs:19-64): [INPUT_FILE|-] [OUTPUT_FILE|-] [-d|--decompress], stdin/stdout by default,
// the file streamed through `compress` / `decompress`.  Additions: -B4..-B7 (block size ID),
// -BI / -BD (independent / linked blocks), --block-checksum, --content-checksum, --batch MiB (gather that much input per
// GPU call; 0 = the reference's 16 KiB-slice conduit verbatim).  Frames are standard LZ4 frames: interchangeable with
// the `lz4` CLI in both directions.  --index: the frame's block list follows it as a skippable frame (lz4f_mi355x_appendBlockList's
// format; other readers skip it, the device decoder finds the blocks through it).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/lz4f_mi355x.h"            // the C ABI only: this program is an ordinary client of the library

static void usage(FILE* f)
{
    fputs("Usage: mi355x-lz4c [INPUT_FILE] [OUTPUT-FILE] [-d|--decompress] [-B4|-B5|-B6|-B7] [-BI|-BD]\n"
          "                   [--block-checksum] [--content-checksum] [--batch MiB] [--index]\n"
          "  Compress or decompress .lz4 files\n", f);
}

struct Io { FILE* in; FILE* out; std::vector<uint8_t> buf; bool write_failed; };
// source: like Data.Conduit.Binary's sourceHandle / sourceFile (32 KiB chunks; larger when batching)
static size_t await_cb(void* user, const void** data)
{
    Io* io = (Io*)user;
    const size_t n = fread(io->buf.data(), 1, io->buf.size(), io->in);
    *data = n ? io->buf.data() : nullptr;
    return n;
}
static void yield_cb(void* user, const void* data, size_t size)
{
    Io* io = (Io*)user;
    if (size && fwrite(data, 1, size, io->out) != size) io->write_failed = true;
}

int main(int argc, char** argv)
{
    std::vector<std::string> pos;
    bool dec = false, listed = false;
    size_t batch = (size_t)64 << 20;
    LZ4F_preferences_t prefs; memset(&prefs, 0, sizeof(prefs));     // = lz4DefaultPreferences (Conduit.hsc:248-263): 64 KiB linked blocks, no checksums
    prefs.frameInfo.blockSizeID = LZ4F_max64KB;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "-d" || a == "--decompress") dec = true;
        else if (a == "-h" || a == "--help") { usage(stdout); return 0; }
        else if (a.size() == 3 && a[0] == '-' && a[1] == 'B' && a[2] >= '4' && a[2] <= '7') prefs.frameInfo.blockSizeID = (LZ4F_blockSizeID_t)(a[2] - '0');
        else if (a == "-BI") prefs.frameInfo.blockMode = LZ4F_blockIndependent;
        else if (a == "-BD") prefs.frameInfo.blockMode = LZ4F_blockLinked;
        else if (a == "--block-checksum") prefs.frameInfo.blockChecksumFlag = LZ4F_blockChecksumEnabled;
        else if (a == "--content-checksum") prefs.frameInfo.contentChecksumFlag = LZ4F_contentChecksumEnabled;
        else if (a == "--index") listed = true;
        else if (a == "--batch" && i + 1 < argc) batch = (size_t)strtoull(argv[++i], nullptr, 10) << 20;
        else if (a == "-" || a[0] != '-') pos.push_back(a);
        else { usage(stderr); return 2; }
    }
    if (pos.size() > 2) { usage(stderr); return 2; }
    Io io; io.in = stdin; io.out = stdout; io.write_failed = false;
    if (pos.size() >= 1 && pos[0] != "-") { io.in = fopen(pos[0].c_str(), "rb"); if (!io.in) { perror(pos[0].c_str()); return 1; } }
    if (pos.size() >= 2 && pos[1] != "-") { io.out = fopen(pos[1].c_str(), "wb"); if (!io.out) { perror(pos[1].c_str()); return 1; } }
    io.buf.resize(batch ? (size_t)4 << 20 : 32752);

    char err[512]; err[0] = 0;
    int rc;
    if (dec) rc = batch ? lz4f_mi355x_conduit_decompress_batched(await_cb, yield_cb, &io, err, sizeof(err))
                        : lz4f_mi355x_conduit_decompress(await_cb, yield_cb, &io, err, sizeof(err));
    else if (listed && !batch) { fputs("mi355x-lz4c: --index needs --batch > 0\n", stderr); return 2; }
    else     rc = listed ? lz4f_mi355x_conduit_compress_batched_listed(batch, &prefs, await_cb, yield_cb, &io, err, sizeof(err))
               : batch ? lz4f_mi355x_conduit_compress_batched(batch, &prefs, await_cb, yield_cb, &io, err, sizeof(err))
                        : lz4f_mi355x_conduit_compress(0, &prefs, await_cb, yield_cb, &io, err, sizeof(err));
    if (rc != 0) { fprintf(stderr, "mi355x-lz4c: %s\n", err[0] ? err : "failed"); return 1; }
    if (io.write_failed || fflush(io.out) != 0) { fputs("mi355x-lz4c: write failed\n", stderr); return 1; }
    if (io.out != stdout) fclose(io.out);
    if (io.in != stdin) fclose(io.in);
    return 0;
}
