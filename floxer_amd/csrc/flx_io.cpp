// SAM / BAM record writer with the reference's record conventions (output.cpp:49-108, 197-212; seqan3::sam_file_output):
//   header @HD VN:1.6 + one @SQ per reference; MAPQ 255; RNEXT * / PNEXT 0 / TLEN 0; NM tag on mapped records;
//   the primary record carries the forward read + qualities, secondary records have empty SEQ/QUAL;
//   unmapped records: flag 4, RNAME *, no tags. Format chosen by extension (output.hpp:33-38). BGZF via zlib.
#include <zlib.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "flx_internal.hpp"

using namespace flx;

struct flx_sam_writer {
    FILE* f = nullptr;
    bool bam = false;
    std::vector<std::string> ref_ids;
    std::vector<uint64_t> ref_lens;
    std::vector<uint8_t> block;      // pending uncompressed BGZF payload
    bool failed = false;
};

namespace {

constexpr size_t BGZF_BLOCK = 0xff00;

bool bgzf_flush_block(flx_sam_writer* w, const uint8_t* data, size_t len) {
    uint8_t out[0x10000 + 64];
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    zs.next_in = const_cast<Bytef*>(data);
    zs.avail_in = (uInt)len;
    zs.next_out = out + 18;
    zs.avail_out = sizeof(out) - 18 - 8;
    int const rc = deflate(&zs, Z_FINISH);
    size_t const clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    size_t const bsize = clen + 18 + 8;
    static const uint8_t hdr[16] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 'B', 'C', 2, 0};
    memcpy(out, hdr, 16);
    out[16] = (uint8_t)((bsize - 1) & 0xff);
    out[17] = (uint8_t)((bsize - 1) >> 8);
    uint32_t const crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)len);
    uint32_t const isize = (uint32_t)len;
    memcpy(out + 18 + clen, &crc, 4);
    memcpy(out + 18 + clen + 4, &isize, 4);
    return fwrite(out, 1, bsize, w->f) == bsize;
}

bool emit(flx_sam_writer* w, const void* data, size_t len) {
    if (!w->bam) return fwrite(data, 1, len, w->f) == len;
    const uint8_t* p = (const uint8_t*)data;
    while (len) {
        size_t const take = std::min(len, BGZF_BLOCK - w->block.size());
        w->block.insert(w->block.end(), p, p + take);
        p += take;
        len -= take;
        if (w->block.size() == BGZF_BLOCK) {
            if (!bgzf_flush_block(w, w->block.data(), w->block.size())) return false;
            w->block.clear();
        }
    }
    return true;
}

template <class T> void put(std::vector<uint8_t>& v, T x) { const uint8_t* p = (const uint8_t*)&x; v.insert(v.end(), p, p + sizeof(T)); }

int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

std::string header_text(flx_sam_writer const* w) {
    std::string h = "@HD\tVN:1.6\n";
    for (size_t i = 0; i < w->ref_ids.size(); ++i) h += "@SQ\tSN:" + w->ref_ids[i] + "\tLN:" + std::to_string(w->ref_lens[i]) + "\n";
    return h;
}

}  // namespace

extern "C" int flx_sam_open(const char* path, const char* const* ref_ids, const uint64_t* ref_lens, uint32_t n_refs, flx_sam_writer** out) {
    if (!path || !out || (n_refs && (!ref_ids || !ref_lens))) { set_error("flx_sam_open: null argument"); return FLX_ERR_INVALID; }
    std::string const p(path);
    bool bam;
    if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".bam") == 0) bam = true;
    else if (p.size() >= 4 && p.compare(p.size() - 4, 4, ".sam") == 0) bam = false;
    else { set_error("output file must end in .sam or .bam (floxer_cli.cpp:258)"); return FLX_ERR_INVALID; }
    FILE* f = fopen(path, "wb");
    if (!f) { set_error(std::string("cannot open output file: ") + path); return FLX_ERR_IO; }
    auto* w = new flx_sam_writer();
    w->f = f;
    w->bam = bam;
    for (uint32_t i = 0; i < n_refs; ++i) { w->ref_ids.emplace_back(ref_ids[i]); w->ref_lens.push_back(ref_lens[i]); }
    std::string const text = header_text(w);
    bool ok;
    if (!bam) ok = emit(w, text.data(), text.size());
    else {
        std::vector<uint8_t> h;
        h.insert(h.end(), {'B', 'A', 'M', 1});
        put<int32_t>(h, (int32_t)text.size());
        h.insert(h.end(), text.begin(), text.end());
        put<int32_t>(h, (int32_t)n_refs);
        for (uint32_t i = 0; i < n_refs; ++i) {
            put<int32_t>(h, (int32_t)w->ref_ids[i].size() + 1);
            h.insert(h.end(), w->ref_ids[i].begin(), w->ref_ids[i].end());
            h.push_back(0);
            put<int32_t>(h, (int32_t)w->ref_lens[i]);
        }
        ok = emit(w, h.data(), h.size());
    }
    if (!ok) { fclose(f); delete w; set_error("write error"); return FLX_ERR_IO; }
    *out = w;
    return FLX_OK;
}

extern "C" int flx_sam_write(flx_sam_writer* w, const char* const* read_ids, const uint8_t* read_pool, const uint64_t* read_offsets,
                             const char* const* quals, const flx_record* records, uint64_t n_records, const uint32_t* cigar_words) {
    if (!w || (n_records && (!records || !read_ids || !read_pool || !read_offsets))) { set_error("flx_sam_write: null argument"); return FLX_ERR_INVALID; }
    static const char ops[] = "MIDNSHP=X";
    std::string line;
    std::vector<uint8_t> rec;
    for (uint64_t i = 0; i < n_records; ++i) {
        flx_record const& r = records[i];
        const char* id = read_ids[r.read_index];
        bool const unmapped = (r.flag & 4u) != 0;
        bool const with_seq = unmapped || !(r.flag & 256u);        // primary or unmapped carry SEQ/QUAL (output.cpp:69-72, 97-105)
        const uint8_t* seq = read_pool + read_offsets[r.read_index];
        uint64_t const slen = with_seq ? read_offsets[r.read_index + 1] - read_offsets[r.read_index] : 0;
        const char* qual = (with_seq && quals) ? quals[r.read_index] : nullptr;
        const uint32_t* cig = cigar_words ? cigar_words + r.cigar_offset : nullptr;
        if (!w->bam) {
            line.clear();
            line += id; line += '\t'; line += std::to_string(r.flag); line += '\t';
            line += unmapped ? "*" : w->ref_ids[r.reference_id]; line += '\t';
            line += std::to_string((int64_t)r.position + 1);       // seqan3 writes ref_offset + 1 (also for the 0 floxer passes when unmapped)
            line += "\t255\t";
            if (r.cigar_length == 0) line += '*';
            else for (uint32_t c = 0; c < r.cigar_length; ++c) { line += std::to_string(cig[c] >> 4); line += ops[cig[c] & 15]; }
            line += "\t*\t0\t0\t";
            if (slen == 0) line += '*';
            else for (uint64_t b = 0; b < slen; ++b) line += rank_to_char(seq[b]);
            line += '\t';
            if (slen == 0 || !qual || !*qual) line += '*';
            else line.append(qual, slen);
            if (!unmapped) { line += "\tNM:i:"; line += std::to_string(r.num_errors); }
            line += '\n';
            if (!emit(w, line.data(), line.size())) { w->failed = true; break; }
        } else {
            rec.clear();
            size_t const l_name = strlen(id) + 1;
            int64_t ref_span = 0;
            for (uint32_t c = 0; c < r.cigar_length; ++c) { uint32_t const op = cig[c] & 15; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_span += cig[c] >> 4; }
            int32_t const pos = r.position;
            put<int32_t>(rec, 0);                                  // block_size, patched below
            put<int32_t>(rec, unmapped ? -1 : r.reference_id);
            put<int32_t>(rec, pos);
            rec.push_back((uint8_t)l_name);
            rec.push_back(255);
            put<uint16_t>(rec, (uint16_t)reg2bin(pos, pos + (ref_span ? ref_span : 1)));
            put<uint16_t>(rec, (uint16_t)r.cigar_length);
            put<uint16_t>(rec, (uint16_t)r.flag);
            put<int32_t>(rec, (int32_t)slen);
            put<int32_t>(rec, -1);
            put<int32_t>(rec, -1);
            put<int32_t>(rec, 0);
            rec.insert(rec.end(), id, id + l_name);
            for (uint32_t c = 0; c < r.cigar_length; ++c) put<uint32_t>(rec, cig[c]);
            static const uint8_t nib[6] = {15, 1, 2, 4, 8, 15};   // =ACMGRSVTWYHKDBN codes for $ACGTN
            for (uint64_t b = 0; b < slen; b += 2) {
                uint8_t const hi = nib[seq[b] < 6 ? seq[b] : 5], lo = b + 1 < slen ? nib[seq[b + 1] < 6 ? seq[b + 1] : 5] : 0;
                rec.push_back((uint8_t)(hi << 4 | lo));
            }
            for (uint64_t b = 0; b < slen; ++b) rec.push_back(qual && *qual ? (uint8_t)(qual[b] - 33) : 0xff);
            if (!unmapped) {
                rec.push_back('N'); rec.push_back('M');
                if (r.num_errors < 256) { rec.push_back('C'); rec.push_back((uint8_t)r.num_errors); }
                else if (r.num_errors < 65536) { rec.push_back('S'); put<uint16_t>(rec, (uint16_t)r.num_errors); }
                else { rec.push_back('I'); put<uint32_t>(rec, r.num_errors); }
            }
            int32_t const bs = (int32_t)rec.size() - 4;
            memcpy(rec.data(), &bs, 4);
            if (!emit(w, rec.data(), rec.size())) { w->failed = true; break; }
        }
    }
    if (w->failed) { set_error("write error on the alignment output"); return FLX_ERR_IO; }
    return FLX_OK;
}

extern "C" int flx_sam_close(flx_sam_writer* w) {
    if (!w) return FLX_OK;
    bool ok = !w->failed;
    if (w->bam) {
        if (!w->block.empty()) ok = bgzf_flush_block(w, w->block.data(), w->block.size()) && ok;
        ok = bgzf_flush_block(w, nullptr, 0) && ok;               // BGZF EOF marker block
    }
    ok = (fclose(w->f) == 0) && ok;
    delete w;
    if (!ok) { set_error("write error while closing the alignment output"); return FLX_ERR_IO; }
    return FLX_OK;
}
