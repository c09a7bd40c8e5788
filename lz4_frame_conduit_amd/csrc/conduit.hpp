// conduit.hpp -- C++ mirror of Codec.Compression.LZ4.Conduit (see conduit.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/lz4f_mi355x.h"

namespace lz4f {
namespace conduit {

struct Slice { const uint8_t* data; size_t size; };          // a strict ByteString
using Await = std::function<bool(Slice&)>;                    // false = upstream finished (await == Nothing)
using Yield = std::function<void(Slice)>;                     // the bytes are only valid during the call (packCStringLen copies)

LZ4F_preferences_t lz4DefaultPreferences();                                                   // Conduit.hsc:248-263
std::vector<Slice> bsChunksOf(size_t chunkSize, Slice bs);                                     // Conduit.hsc:428-433
void compress(const Await&, const Yield&);                                                     // Conduit.hsc:336-337
void compressYieldImmediately(const LZ4F_preferences_t* prefs, const Await&, const Yield&);    // Conduit.hsc:364-425
void compressWithOutBufferSize(size_t bufferSize, const LZ4F_preferences_t* prefs, const Await&, const Yield&);   // Conduit.hsc:457-533
void decompress(const Await&, const Yield&);                                                   // Conduit.hsc:598-701
// additions: preferences as a parameter, and batched drivers for the GPU
void compressWithPreferences(const LZ4F_preferences_t& prefs, const Await&, const Yield&);
void compressBatched(size_t batchBytes, const LZ4F_preferences_t* prefs, const Await&, const Yield&, bool blockList = false);
void decompressBatched(const Await&, const Yield&, size_t batchBytes = (size_t)256 << 20);     // bounded memory: a batch of whole blocks at a time

}  // namespace conduit
}  // namespace lz4f
