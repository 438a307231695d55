// TEST INFRASTRUCTURE -- a stand-in for librocdecode.so, so that csrc/rva_decode.hip EXECUTES on boxes that do not carry
// the real library (neither the build container nor this project's GPU boxes do).  It exports the entry points
// rocdec_api() resolves (rva_decode.hip:64-74) with the signatures of the rocDecode 0.10 headers in the ROCm image and
// honours their calling contract: sequence callback on a (new) sequence header, decode callback per picture in decode
// order, display callback in display order (one picture late, so that the end-of-stream flush matters), pictures stay
// owned by the client until rocDecParserMarkFrameForReuse.  It pins NOTHING about what a VCN decoder produces: the
// "bitstream" is a toy format carried in Annex-B NAL units (below) and the "decoded" pictures are a closed-form pattern.
//
// Toy stream (every number is three bytes 0x40 | 6 bits, so no start code can appear inside a NAL unit):
//   NAL type 7 ("SPS"):   header byte 0x67, then coded_w, coded_h, left, top, right, bottom
//   NAL type 5 / 1 (slice): header byte 0x65 / 0x41, then 0x80 | .. (first_mb_in_slice = 0), then the frame number
// Picture f of a sequence, coded coordinates (x, y):  Y = (3x + 5y + 7f) & 0xff;  U = (x/2 + 3(y/2) + 11f) & 0xff at byte
// 2(x/2), V = (5(x/2) + (y/2) + 13f) & 0xff at byte 2(x/2) + 1 of the interleaved plane.  tests/decode_worker.py restates it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include <rocprofiler-sdk/rocdecode/details/rocdecode.h>
#include <rocprofiler-sdk/rocdecode/details/rocparser.h>

namespace {

struct MockDecoder {
    uint32_t w = 0, h = 0, pitch = 0, n = 0;
    std::vector<uint8_t *> surf;          // device: Y plane (h rows) followed by the UV plane (h / 2 rows), same pitch
    std::vector<int> frame_of;            // frame number "decoded" into each surface
};

struct MockParser {
    RocdecParserParams pp{};
    RocdecVideoFormat fmt{};
    bool have_seq = false;
    int n_surf = 0;
    std::vector<char> busy;               // handed to the client by a display callback, not yet marked for reuse
    int held_pic = -1;                    // decoded, display pending (one picture of delay)
    RocdecTimeStamp held_pts = 0;
    int pending_frame = 0;                // frame number of the picture being decoded (parser -> decoder, same library)
};

MockParser *g_decoding = nullptr;         // the parser whose decode callback is running (for rocDecDecodeFrame)

unsigned num3(const uint8_t *p) { return ((p[0] & 0x3Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu); }

bool display(MockParser *ps, int pic, RocdecTimeStamp pts)
{
    RocdecParserDispInfo di{};
    di.picture_index = pic; di.progressive_frame = 1; di.pts = pts;
    ps->busy[pic] = 1;
    return ps->pp.pfn_display_picture(ps->pp.user_data, &di) != 0;
}

}  // namespace

extern "C" {

rocDecStatus ROCDECAPI rocDecCreateVideoParser(RocdecVideoParser *out, RocdecParserParams *pp)
{
    if (!out || !pp || !pp->pfn_sequence_callback || !pp->pfn_decode_picture || !pp->pfn_display_picture) return ROCDEC_INVALID_PARAMETER;
    MockParser *ps = new MockParser();
    ps->pp = *pp;
    *out = ps;
    return ROCDEC_SUCCESS;
}

rocDecStatus ROCDECAPI rocDecDestroyVideoParser(RocdecVideoParser h)
{
    delete static_cast<MockParser *>(h);
    return ROCDEC_SUCCESS;
}

rocDecStatus ROCDECAPI rocDecParserMarkFrameForReuse(RocdecVideoParser h, int pic)
{
    MockParser *ps = static_cast<MockParser *>(h);
    if (!ps || pic < 0 || pic >= ps->n_surf || !ps->busy[pic]) return ROCDEC_INVALID_PARAMETER;
    ps->busy[pic] = 0;
    return ROCDEC_SUCCESS;
}

rocDecStatus ROCDECAPI rocDecParseVideoData(RocdecVideoParser h, RocdecSourceDataPacket *pkt)
{
    MockParser *ps = static_cast<MockParser *>(h);
    if (!ps || !pkt) return ROCDEC_INVALID_PARAMETER;
    const uint8_t *p = pkt->payload;
    const size_t n = pkt->payload_size;
    size_t i = 0;
    while (p && i + 3 < n) {
        if (!(p[i] == 0 && p[i + 1] == 0 && p[i + 2] == 1)) { ++i; continue; }
        const uint8_t *nal = p + i + 3;
        const size_t left = n - (i + 3);
        const int type = nal[0] & 0x1F;
        if (type == 7 && left >= 19) {
            RocdecVideoFormat f{};
            f.codec = ps->pp.codec_type;
            f.frame_rate.numerator = 30; f.frame_rate.denominator = 1;
            f.progressive_sequence = 1;
            f.min_num_decode_surfaces = 6;
            f.coded_width = num3(nal + 1); f.coded_height = num3(nal + 4);
            f.display_area.left = (int)num3(nal + 7); f.display_area.top = (int)num3(nal + 10);
            f.display_area.right = (int)num3(nal + 13); f.display_area.bottom = (int)num3(nal + 16);
            f.chroma_format = rocDecVideoChromaFormat_420;
            if (!ps->have_seq || std::memcmp(&f, &ps->fmt, sizeof f) != 0) {
                if (ps->held_pic >= 0) {                       // a new sequence flushes the display queue of the old one
                    display(ps, ps->held_pic, ps->held_pts);
                    ps->held_pic = -1;
                }
                const int ns = ps->pp.pfn_sequence_callback(ps->pp.user_data, &f);
                if (ns <= 0) return ROCDEC_RUNTIME_ERROR;
                ps->fmt = f; ps->have_seq = true;
                ps->n_surf = ns > 1 ? ns : (int)ps->pp.max_num_decode_surfaces;
                ps->busy.assign(ps->n_surf, 0);
            }
            i += 3 + 19;
            continue;
        }
        if ((type == 5 || type == 1) && left >= 5) {
            if (!ps->have_seq) return ROCDEC_RUNTIME_ERROR;
            int pic = -1;
            for (int k = 0; k < ps->n_surf; ++k)
                if (!ps->busy[k] && k != ps->held_pic) { pic = k; break; }
            if (pic < 0) return ROCDEC_OUTOF_MEMORY;           // the client holds every surface: nothing to decode into
            RocdecPicParams prm{};
            prm.pic_width = (int)ps->fmt.coded_width; prm.pic_height = (int)ps->fmt.coded_height;
            prm.curr_pic_idx = pic;
            ps->pending_frame = (int)num3(nal + 2);
            g_decoding = ps;
            const int ok = ps->pp.pfn_decode_picture(ps->pp.user_data, &prm);
            g_decoding = nullptr;
            if (!ok) return ROCDEC_RUNTIME_ERROR;
            if (ps->held_pic >= 0 && !display(ps, ps->held_pic, ps->held_pts)) return ROCDEC_RUNTIME_ERROR;
            ps->held_pic = pic;
            ps->held_pts = (pkt->flags & ROCDEC_PKT_TIMESTAMP) ? pkt->pts : 0;
            i += 3 + 5;
            continue;
        }
        i += 3;
    }
    if (pkt->flags & ROCDEC_PKT_ENDOFSTREAM) {
        if (ps->held_pic >= 0) {
            display(ps, ps->held_pic, ps->held_pts);
            ps->held_pic = -1;
        }
        if (pkt->flags & ROCDEC_PKT_NOTIFY_EOS) ps->pp.pfn_display_picture(ps->pp.user_data, nullptr);
    }
    return ROCDEC_SUCCESS;
}

rocDecStatus ROCDECAPI rocDecGetDecoderCaps(RocdecDecodeCaps *c)
{
    if (!c) return ROCDEC_INVALID_PARAMETER;
    c->is_supported = (c->codec_type == rocDecVideoCodec_AVC || c->codec_type == rocDecVideoCodec_HEVC) &&
                      c->chroma_format == rocDecVideoChromaFormat_420 && c->bit_depth_minus_8 == 0;
    c->num_decoders = 1;
    c->output_format_mask = 1u << rocDecVideoSurfaceFormat_NV12;
    c->max_width = 4096; c->max_height = 2304; c->min_width = 16; c->min_height = 16;     // a 4096 x 2304 "VCN"
    return ROCDEC_SUCCESS;
}

rocDecStatus ROCDECAPI rocDecCreateDecoder(rocDecDecoderHandle *out, RocDecoderCreateInfo *ci)
{
    if (!out || !ci || ci->output_format != rocDecVideoSurfaceFormat_NV12 || !ci->width || !ci->height || !ci->num_decode_surfaces)
        return ROCDEC_INVALID_PARAMETER;
    if (hipSetDevice(ci->device_id) != hipSuccess) return ROCDEC_DEVICE_INVALID;
    MockDecoder *d = new MockDecoder();
    d->w = ci->width; d->h = ci->height; d->n = ci->num_decode_surfaces;
    d->pitch = (d->w + 255u) & ~255u;
    d->frame_of.assign(d->n, -1);
    for (uint32_t k = 0; k < d->n; ++k) {
        uint8_t *p = nullptr;
        if (hipMalloc(&p, (size_t)d->pitch * (d->h + d->h / 2)) != hipSuccess) { delete d; return ROCDEC_OUTOF_MEMORY; }
        d->surf.push_back(p);
    }
    *out = d;
    return ROCDEC_SUCCESS;
}

rocDecStatus ROCDECAPI rocDecDestroyDecoder(rocDecDecoderHandle h)
{
    MockDecoder *d = static_cast<MockDecoder *>(h);
    if (!d) return ROCDEC_INVALID_PARAMETER;
    (void)hipDeviceSynchronize();
    for (uint8_t *p : d->surf) (void)hipFree(p);
    delete d;
    return ROCDEC_SUCCESS;
}

rocDecStatus ROCDECAPI rocDecDecodeFrame(rocDecDecoderHandle h, RocdecPicParams *prm)
{
    MockDecoder *d = static_cast<MockDecoder *>(h);
    if (!d || !prm || prm->curr_pic_idx < 0 || (uint32_t)prm->curr_pic_idx >= d->n || !g_decoding) return ROCDEC_INVALID_PARAMETER;
    if ((uint32_t)prm->pic_width != d->w || (uint32_t)prm->pic_height != d->h) return ROCDEC_INVALID_PARAMETER;
    const int f = g_decoding->pending_frame;
    std::vector<uint8_t> img((size_t)d->pitch * (d->h + d->h / 2), 0);
    for (uint32_t y = 0; y < d->h; ++y)
        for (uint32_t x = 0; x < d->w; ++x) img[(size_t)y * d->pitch + x] = (uint8_t)(3 * x + 5 * y + 7 * f);
    uint8_t *uv = img.data() + (size_t)d->pitch * d->h;
    for (uint32_t y = 0; y < d->h / 2; ++y)
        for (uint32_t x = 0; x < d->w / 2; ++x) {
            uv[(size_t)y * d->pitch + 2 * x] = (uint8_t)(x + 3 * y + 11 * f);
            uv[(size_t)y * d->pitch + 2 * x + 1] = (uint8_t)(5 * x + y + 13 * f);
        }
    if (hipMemcpy(d->surf[prm->curr_pic_idx], img.data(), img.size(), hipMemcpyHostToDevice) != hipSuccess) return ROCDEC_RUNTIME_ERROR;
    d->frame_of[prm->curr_pic_idx] = f;
    return ROCDEC_SUCCESS;
}

rocDecStatus ROCDECAPI rocDecGetVideoFrame(rocDecDecoderHandle h, int pic, void *ptr[3], uint32_t *pitch, RocdecProcParams *)
{
    MockDecoder *d = static_cast<MockDecoder *>(h);
    if (!d || !ptr || !pitch || pic < 0 || (uint32_t)pic >= d->n || d->frame_of[pic] < 0) return ROCDEC_INVALID_PARAMETER;
    ptr[0] = d->surf[pic];
    ptr[1] = d->surf[pic] + (size_t)d->pitch * d->h;
    ptr[2] = nullptr;
    pitch[0] = pitch[1] = d->pitch; pitch[2] = 0;
    return ROCDEC_SUCCESS;
}

const char *ROCDECAPI rocDecGetErrorName(rocDecStatus st)
{
    switch (st) {
    case ROCDEC_SUCCESS: return "ROCDEC_SUCCESS";
    case ROCDEC_DEVICE_INVALID: return "ROCDEC_DEVICE_INVALID";
    case ROCDEC_RUNTIME_ERROR: return "ROCDEC_RUNTIME_ERROR";
    case ROCDEC_OUTOF_MEMORY: return "ROCDEC_OUTOF_MEMORY";
    case ROCDEC_INVALID_PARAMETER: return "ROCDEC_INVALID_PARAMETER";
    case ROCDEC_NOT_SUPPORTED: return "ROCDEC_NOT_SUPPORTED";
    default: return "ROCDEC_UNKNOWN";
    }
}

}  // extern "C"
