// D1: H.264 / H.265 elementary stream -> NV12 surfaces in HBM through rocDecode (VCN), replacing what
// cv2.VideoCapture(url, CAP_FFMPEG).read() does on the CPU in the reference (video_stream.py:76,173: FFmpeg software
// decode + swscale to a host BGR array).  Frames never visit the host: the parser's display callback hands out a
// picture index, rocDecGetVideoFrame maps it as HIP device pointers (Y plane + interleaved UV plane + pitch), and K1
// reads those pointers directly.
//
// librocdecode is loaded with dlopen at run time and its entry points are resolved by name, so librva.so has no link
// dependency on it: on a machine without the library rva_decoder_create() returns RVA_ERR_UNAVAILABLE and everything
// else keeps working (rva_decode_available() is the probe).  Compiled against the rocDecode 0.10 headers that ship in
// the ROCm image (rocprofiler-sdk/rocdecode/details).  NOTE: neither the build container nor the GPU boxes of this
// project carry librocdecode.so.  What executes this translation unit there is a TEST DOUBLE (tests/mock_rocdecode/, named
// through RVA_ROCDECODE_LIB): it drives the callbacks, the display queue, the display-area crop, the hold / release of
// mapped pictures, the end-of-stream flush and a mid-stream change of picture size, and pins nothing about VCN pixels.
#include <dlfcn.h>

#include <deque>
#include <mutex>

#include <rocprofiler-sdk/rocdecode/details/rocdecode.h>
#include <rocprofiler-sdk/rocdecode/details/rocparser.h>

#include "rva_internal.h"

namespace {

struct RocDecApi {
    void *lib = nullptr;
    rocDecStatus (*CreateVideoParser)(RocdecVideoParser *, RocdecParserParams *) = nullptr;
    rocDecStatus (*ParseVideoData)(RocdecVideoParser, RocdecSourceDataPacket *) = nullptr;
    rocDecStatus (*DestroyVideoParser)(RocdecVideoParser) = nullptr;
    rocDecStatus (*ParserMarkFrameForReuse)(RocdecVideoParser, int) = nullptr;   // optional (newer releases)
    rocDecStatus (*CreateDecoder)(rocDecDecoderHandle *, RocDecoderCreateInfo *) = nullptr;
    rocDecStatus (*DestroyDecoder)(rocDecDecoderHandle) = nullptr;
    rocDecStatus (*GetDecoderCaps)(RocdecDecodeCaps *) = nullptr;                 // optional
    rocDecStatus (*DecodeFrame)(rocDecDecoderHandle, RocdecPicParams *) = nullptr;
    rocDecStatus (*GetVideoFrame)(rocDecDecoderHandle, int, void *[3], uint32_t *, RocdecProcParams *) = nullptr;
    rocDecStatus (*ReconfigureDecoder)(rocDecDecoderHandle, RocdecReconfigureDecoderInfo *) = nullptr;   // optional
    const char *(*GetErrorName)(rocDecStatus) = nullptr;                         // optional
    std::string where;
};

// one load per process; never unloaded (decoder threads of the library may outlive a session)
RocDecApi *rocdec_api(std::string *why)
{
    static std::mutex mu;
    static RocDecApi api;
    static bool tried = false;
    static std::string failure;
    std::lock_guard<std::mutex> g(mu);
    if (!tried) {
        tried = true;
        // RVA_ROCDECODE_LIB: an explicit library path tried first (a rocDecode build outside the default search path)
        const char *names[] = {getenv("RVA_ROCDECODE_LIB"), "librocdecode.so", "librocdecode.so.1", "librocdecode.so.0",
                               "/opt/rocm/lib/librocdecode.so"};
        for (const char *nm : names) {
            if (!nm || !*nm) continue;
            api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) { api.where = nm; break; }
        }
        if (!api.lib) {
            failure = "librocdecode.so not found (dlopen failed)";
        } else {
            bool ok = true;
            auto need = [&](auto &fn, const char *sym, bool required) {
                fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.lib, sym));
                if (!fn && required) { ok = false; failure = std::string(api.where) + ": missing symbol " + sym; }
            };
            need(api.CreateVideoParser, "rocDecCreateVideoParser", true);
            need(api.ParseVideoData, "rocDecParseVideoData", true);
            need(api.DestroyVideoParser, "rocDecDestroyVideoParser", true);
            need(api.ParserMarkFrameForReuse, "rocDecParserMarkFrameForReuse", false);
            need(api.CreateDecoder, "rocDecCreateDecoder", true);
            need(api.DestroyDecoder, "rocDecDestroyDecoder", true);
            need(api.GetDecoderCaps, "rocDecGetDecoderCaps", false);
            need(api.DecodeFrame, "rocDecDecodeFrame", true);
            need(api.GetVideoFrame, "rocDecGetVideoFrame", true);
            need(api.ReconfigureDecoder, "rocDecReconfigureDecoder", false);
            need(api.GetErrorName, "rocDecGetErrorName", false);
            if (!ok) { dlclose(api.lib); api.lib = nullptr; }
        }
    }
    if (!api.lib) { if (why) *why = failure; return nullptr; }
    return &api;
}

}  // namespace

struct rva_decoder {
    rva_ctx *ctx = nullptr;
    RocDecApi *api = nullptr;
    rocDecVideoCodec codec = rocDecVideoCodec_HEVC;
    RocdecVideoParser parser = nullptr;
    rocDecDecoderHandle dec = nullptr;   // decoder of the current sequence (= gens.back().dec)
    int want_surfaces = 0;
    // sequence state (sequence callback)
    uint32_t coded_w = 0, coded_h = 0;
    int disp_w = 0, disp_h = 0, crop_left = 0, crop_top = 0;
    uint32_t surfaces = 0;
    // A decoder lives as long as pictures of it are queued for display or mapped by the caller: a change of picture size
    // mid-stream starts a new generation, the old decoder is destroyed when its last picture has been released (the
    // parser flushes the old sequence's pictures through the display callback right before it announces the new one).
    struct Gen { int id; rocDecDecoderHandle dec; int disp_w, disp_h, crop_left, crop_top; int outstanding; };
    std::deque<Gen> gens;
    int next_gen = 0;
    // display queue (display callback -> rva_decoder_next_frame)
    struct Ready { int pic; int64_t pts; int gen; };
    std::deque<Ready> ready;
    bool eos_sent = false;
    std::string cb_error;           // first error raised inside a callback (callbacks cannot return text)
    uint64_t frames_out = 0;
};

namespace {

const char *dec_err(rva_decoder *d, rocDecStatus st)
{
    return d->api->GetErrorName ? d->api->GetErrorName(st) : "rocDecode error";
}

int ROCDECAPI on_sequence(void *user, RocdecVideoFormat *fmt)
{
    rva_decoder *d = static_cast<rva_decoder *>(user);
    if (!fmt) return 0;
    if (fmt->chroma_format != rocDecVideoChromaFormat_420 || fmt->bit_depth_luma_minus8 != 0 || fmt->bit_depth_chroma_minus8 != 0) {
        d->cb_error = "only 8-bit 4:2:0 streams map onto the NV12 pre-process (K1)";
        return 0;
    }
    const uint32_t n = fmt->min_num_decode_surfaces > (uint32_t)d->want_surfaces ? fmt->min_num_decode_surfaces : (uint32_t)d->want_surfaces;
    const int dw = fmt->display_area.right - fmt->display_area.left, dh = fmt->display_area.bottom - fmt->display_area.top;
    if (d->dec && fmt->coded_width == d->coded_w && fmt->coded_height == d->coded_h && n <= d->surfaces) return (int)d->surfaces;
    if (d->dec) {   // resolution change mid-stream: a decoder of the new size; the old one goes when its pictures are back
        d->dec = nullptr;
        if (!d->gens.empty() && d->gens.back().outstanding == 0) {
            d->api->DestroyDecoder(d->gens.back().dec);
            d->gens.pop_back();
        }
    }
    if (d->api->GetDecoderCaps) {
        RocdecDecodeCaps caps{};
        caps.device_id = (uint8_t)d->ctx->device;
        caps.codec_type = fmt->codec;
        caps.chroma_format = fmt->chroma_format;
        caps.bit_depth_minus_8 = 0;
        if (d->api->GetDecoderCaps(&caps) == ROCDEC_SUCCESS &&
            (!caps.is_supported || fmt->coded_width > caps.max_width || fmt->coded_height > caps.max_height)) {
            d->cb_error = "the VCN decoder of this GPU does not support this codec / picture size";
            return 0;
        }
    }
    RocDecoderCreateInfo ci{};
    ci.device_id = (uint8_t)d->ctx->device;
    ci.width = fmt->coded_width;
    ci.height = fmt->coded_height;
    ci.num_decode_surfaces = n;
    ci.codec_type = fmt->codec;
    ci.chroma_format = fmt->chroma_format;
    ci.bit_depth_minus_8 = 0;
    ci.intra_decode_only = 0;
    ci.max_width = fmt->coded_width;
    ci.max_height = fmt->coded_height;
    ci.display_rect.left = (int16_t)fmt->display_area.left;
    ci.display_rect.top = (int16_t)fmt->display_area.top;
    ci.display_rect.right = (int16_t)fmt->display_area.right;
    ci.display_rect.bottom = (int16_t)fmt->display_area.bottom;
    ci.output_format = rocDecVideoSurfaceFormat_NV12;
    ci.target_width = (uint32_t)((dw + 1) & ~1);
    ci.target_height = (uint32_t)((dh + 1) & ~1);
    ci.num_output_surfaces = 1;
    rocDecStatus st = d->api->CreateDecoder(&d->dec, &ci);
    if (st != ROCDEC_SUCCESS) {
        d->dec = nullptr;
        d->cb_error = std::string("rocDecCreateDecoder failed: ") + dec_err(d, st);
        return 0;
    }
    d->coded_w = fmt->coded_width; d->coded_h = fmt->coded_height;
    d->disp_w = dw; d->disp_h = dh; d->crop_left = fmt->display_area.left; d->crop_top = fmt->display_area.top;
    d->surfaces = n;
    d->gens.push_back({d->next_gen++, d->dec, d->disp_w, d->disp_h, d->crop_left, d->crop_top, 0});
    return (int)n;       // > 1: the parser adopts this as its DPB size
}

int ROCDECAPI on_decode(void *user, RocdecPicParams *pic)
{
    rva_decoder *d = static_cast<rva_decoder *>(user);
    if (!d->dec || !pic) { if (d->cb_error.empty()) d->cb_error = "picture before a valid sequence header"; return 0; }
    rocDecStatus st = d->api->DecodeFrame(d->dec, pic);
    if (st != ROCDEC_SUCCESS) { d->cb_error = std::string("rocDecDecodeFrame failed: ") + dec_err(d, st); return 0; }
    return 1;
}

int ROCDECAPI on_display(void *user, RocdecParserDispInfo *info)
{
    rva_decoder *d = static_cast<rva_decoder *>(user);
    if (!info) return 1;     // end-of-stream marker (ROCDEC_PKT_NOTIFY_EOS)
    if (d->gens.empty() || info->picture_index < 0 || info->picture_index >= 1024) {
        if (d->cb_error.empty()) d->cb_error = "display callback without a decoder / with a picture index out of range";
        return 0;
    }
    d->gens.back().outstanding += 1;
    d->ready.push_back({info->picture_index, (int64_t)info->pts, d->gens.back().id});
    return 1;
}

}  // namespace

extern "C" {

int rva_decode_available(char *detail, int detail_len)
{
    std::string why;
    RocDecApi *api = rocdec_api(&why);
    if (detail && detail_len > 0) snprintf(detail, detail_len, "%s", api ? (api->where + " ok").c_str() : why.c_str());
    return api ? RVA_OK : RVA_ERR_UNAVAILABLE;
}

int rva_decoder_create(rva_ctx *ctx, int codec, int num_surfaces, rva_decoder **out)
{
    if (!ctx || !out || (codec != RVA_CODEC_H264 && codec != RVA_CODEC_HEVC)) return rva_fail(ctx, RVA_ERR_ARG, "rva_decoder_create: bad argument");
    *out = nullptr;
    std::string why;
    RocDecApi *api = rocdec_api(&why);
    if (!api) return rva_fail(ctx, RVA_ERR_UNAVAILABLE, "rocDecode unavailable: %s", why.c_str());
    RVA_HIP(ctx, hipSetDevice(ctx->device));
    rva_decoder *d = new rva_decoder();
    d->ctx = ctx; d->api = api;
    d->codec = codec == RVA_CODEC_H264 ? rocDecVideoCodec_AVC : rocDecVideoCodec_HEVC;
    d->want_surfaces = num_surfaces > 0 ? num_surfaces : 8;
    RocdecParserParams pp{};
    pp.codec_type = d->codec;
    pp.max_num_decode_surfaces = 1;          // the sequence callback reports the real number
    pp.clock_rate = 0;                       // 10 MHz timestamps
    pp.error_threshold = 0;
    pp.max_display_delay = 0;                // hand frames out as soon as they are displayable (lowest latency)
    pp.user_data = d;
    pp.pfn_sequence_callback = on_sequence;
    pp.pfn_decode_picture = on_decode;
    pp.pfn_display_picture = on_display;
    pp.pfn_get_sei_msg = nullptr;
    rocDecStatus st = api->CreateVideoParser(&d->parser, &pp);
    if (st != ROCDEC_SUCCESS) {
        const char *nm = dec_err(d, st);
        delete d;
        return rva_fail(ctx, RVA_ERR_HIP, "rocDecCreateVideoParser failed: %s", nm);
    }
    *out = d;
    return RVA_OK;
}

void rva_decoder_destroy(rva_decoder *d)
{
    if (!d) return;
    if (d->parser) d->api->DestroyVideoParser(d->parser);
    for (auto &g : d->gens) d->api->DestroyDecoder(g.dec);
    delete d;
}

int rva_decoder_feed(rva_decoder *d, const uint8_t *data, int size, int64_t pts, int end_of_stream)
{
    if (!d || size < 0 || (size > 0 && !data)) return RVA_ERR_ARG;
    if (d->eos_sent) return rva_fail(d->ctx, RVA_ERR_ARG, "rva_decoder_feed: the stream was already ended");
    RocdecSourceDataPacket pkt{};
    pkt.payload = data;
    pkt.payload_size = (uint32_t)size;
    pkt.flags = ROCDEC_PKT_TIMESTAMP | (size > 0 ? ROCDEC_PKT_ENDOFPICTURE : 0);   // one access unit per call
    pkt.pts = (RocdecTimeStamp)pts;
    if (end_of_stream) { pkt.flags |= ROCDEC_PKT_ENDOFSTREAM; d->eos_sent = true; }
    d->cb_error.clear();
    rocDecStatus st = d->api->ParseVideoData(d->parser, &pkt);
    if (!d->cb_error.empty()) return rva_fail(d->ctx, RVA_ERR_HIP, "decode: %s", d->cb_error.c_str());
    if (st != ROCDEC_SUCCESS) return rva_fail(d->ctx, RVA_ERR_HIP, "rocDecParseVideoData failed: %s", dec_err(d, st));
    return RVA_OK;
}

int rva_decoder_next_frame(rva_decoder *d, void **y, void **uv, int32_t *pitch, int32_t *width, int32_t *height,
                           int64_t *pts, int32_t *pic_index)
{
    if (!d || !y || !uv || !pitch || !width || !height || !pic_index) return RVA_ERR_ARG;
    *pic_index = -1;
    if (d->ready.empty()) return RVA_OK;                        // nothing displayable yet: feed more data
    const rva_decoder::Ready r = d->ready.front();
    d->ready.pop_front();
    rva_decoder::Gen *g = nullptr;
    for (auto &x : d->gens) if (x.id == r.gen) g = &x;
    if (!g) return rva_fail(d->ctx, RVA_ERR_HIP, "decode: a queued picture outlived its decoder");
    void *planes[3] = {nullptr, nullptr, nullptr};
    uint32_t pitches[3] = {0, 0, 0};
    RocdecProcParams proc{};
    proc.progressive_frame = 1;
    rocDecStatus st = d->api->GetVideoFrame(g->dec, r.pic, planes, pitches, &proc);   // waits for the picture, maps it for HIP
    if (st != ROCDEC_SUCCESS) return rva_fail(d->ctx, RVA_ERR_HIP, "rocDecGetVideoFrame failed: %s", dec_err(d, st));
    if (!planes[0] || !planes[1] || pitches[0] == 0 || pitches[1] != pitches[0])
        return rva_fail(d->ctx, RVA_ERR_HIP, "rocDecGetVideoFrame returned an unexpected NV12 layout (pitch %u / %u)", pitches[0], pitches[1]);
    // crop to the display area: whole chroma sample pairs only (K1 addresses UV at (y >> 1, (x >> 1) << 1))
    const int cx = g->crop_left & ~1, cy = g->crop_top & ~1;
    *y = (uint8_t *)planes[0] + (size_t)cy * pitches[0] + cx;
    *uv = (uint8_t *)planes[1] + (size_t)(cy >> 1) * pitches[1] + cx;
    *pitch = (int32_t)pitches[0];
    *width = g->disp_w & ~1;
    *height = g->disp_h & ~1;
    if (pts) *pts = r.pts;
    *pic_index = (r.gen << 10) | r.pic;       // the ticket rva_decoder_release takes back: decoder generation + picture
    ++d->frames_out;
    return RVA_OK;
}

int rva_decoder_release(rva_decoder *d, int pic_index)
{
    if (!d || pic_index < 0) return RVA_ERR_ARG;
    const int gen = pic_index >> 10, pic = pic_index & 1023;
    for (size_t i = 0; i < d->gens.size(); ++i) {
        rva_decoder::Gen &g = d->gens[i];
        if (g.id != gen) continue;
        if (g.outstanding <= 0) return rva_fail(d->ctx, RVA_ERR_ARG, "rva_decoder_release: picture %d was not handed out", pic_index);
        g.outstanding -= 1;
        const bool current = g.dec == d->dec;
        if (!current && g.outstanding == 0) {                   // last picture of a retired decoder
            d->api->DestroyDecoder(g.dec);
            d->gens.erase(d->gens.begin() + (long)i);
            return RVA_OK;
        }
        if (current && d->api->ParserMarkFrameForReuse && d->parser) {   // the parser's pool belongs to the current sequence
            rocDecStatus st = d->api->ParserMarkFrameForReuse(d->parser, pic);
            if (st != ROCDEC_SUCCESS) return rva_fail(d->ctx, RVA_ERR_HIP, "rocDecParserMarkFrameForReuse failed: %s", dec_err(d, st));
        }
        return RVA_OK;
    }
    return rva_fail(d->ctx, RVA_ERR_ARG, "rva_decoder_release: picture %d belongs to no live decoder", pic_index);
}

}  // extern "C"
