// Host-side WAV front end of the file-scoring loop: what stands in front of `self.model(wave, lengths)` in
// Nomad.get_embeddings_csv (/root/reference/src/nomad_audio/nomad.py:171-183) is load_processing (nomad.py:192-212):
// torchaudio.load -> fp32 in [-1, 1), first two channels averaged, 16 kHz.  The reference decodes one file per
// iteration on the Python thread; here headers are probed and sample data is converted on plain host threads (no GIL),
// straight into the rows of the pinned staging buffer that nomad_embed_ragged* reads after ONE H2D copy.
// A file at another rate is resampled here too (torchaudio.transforms.Resample defaults: Hann-windowed sinc, width 6,
// rolloff 0.99, nomad.py:203-205).  Decoded values are bit-identical to wavio.read_wav + the two-channel mean; resampled
// values agree with wavio.resample to fp32 summation order (~1e-7).
#pragma once

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>

#include "../../include/nomad_hip.h"

namespace nomad {
namespace wav {

inline uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t le16(const unsigned char* p) { return p[0] | (p[1] << 8); }

inline int bytes_per_sample(int tag, int bits) {
    if (tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) return bits / 8;
    if (tag == 3 && (bits == 32 || bits == 64)) return bits / 8;
    return 0;
}

// Walks the RIFF chunk list like wavio.read_wav: the first 'data' chunk after a 'fmt ' chunk is the audio; a chunk
// that runs past the end of the file is cut at the end of the file; chunks are 2-byte aligned.
inline int probe_one(const char* path, nomad_wav_info* out) {
    memset(out, 0, sizeof(*out));
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return NOMAD_ERR_IO;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return NOMAD_ERR_IO;
    }
    const int64_t fsize = st.st_size;
    unsigned char h[48];
    int rc = NOMAD_ERR_FORMAT;
    bool have_fmt = false;
    if (fsize >= 12 && pread(fd, h, 12, 0) == 12 && !memcmp(h, "RIFF", 4) && !memcmp(h + 8, "WAVE", 4)) {
        int64_t pos = 12;
        while (pos + 8 <= fsize) {
            if (pread(fd, h, 8, pos) != 8) break;
            const int64_t size = le32(h + 4);
            const int64_t avail = size < fsize - (pos + 8) ? size : fsize - (pos + 8);
            if (!memcmp(h, "fmt ", 4)) {
                if (avail < 16) break;
                const int64_t take = avail < 40 ? avail : 40;
                if (pread(fd, h + 8, (size_t)take, pos + 8) != take) break;
                int tag = (int)le16(h + 8);
                if (tag == 0xFFFE && take >= 26) tag = (int)le16(h + 8 + 24);  // WAVE_FORMAT_EXTENSIBLE: sub-format GUID
                out->format_tag = tag;
                out->channels = (int)le16(h + 10);
                out->sample_rate = (int)le32(h + 12);
                out->bits = (int)le16(h + 22);
                have_fmt = true;
            } else if (!memcmp(h, "data", 4)) {
                if (!have_fmt) break;
                const int bps = bytes_per_sample(out->format_tag, out->bits);
                if (bps == 0 || out->channels < 1) break;
                if (out->bits != 24 && avail % bps) break;  // wavio.read_wav rejects a ragged tail (np.frombuffer); 24-bit is cut
                out->data_offset = pos + 8;
                out->frames = avail / bps / out->channels;
                rc = NOMAD_OK;
                break;
            }
            pos += 8 + size + (size & 1);
        }
    }
    close(fd);
    return rc;
}

inline float sample_at(const unsigned char* p, int tag, int bits) {
    if (tag == 1) {
        switch (bits) {
            case 8: return ((float)p[0] - 128.0f) / 128.0f;
            case 16: return (float)(int16_t)le16(p) / 32768.0f;
            case 24: {
                int32_t v = (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16));
                if (v & 0x800000) v -= 0x1000000;
                return (float)v / 8388608.0f;
            }
            default: return (float)((double)(int32_t)le32(p) / 2147483648.0);
        }
    }
    if (bits == 32) {
        float f;
        memcpy(&f, p, 4);
        return f;
    }
    double d;
    memcpy(&d, p, 8);
    return (float)d;
}

// One file -> `frames` mono samples at dst.  raw: a per-thread scratch vector.
inline int read_one(const char* path, const nomad_wav_info& wi, float* dst, std::vector<unsigned char>& raw) {
    const int bps = bytes_per_sample(wi.format_tag, wi.bits);
    if (bps == 0 || wi.channels < 1 || wi.frames < 0) return NOMAD_ERR_FORMAT;
    const int ch = wi.channels;
    const size_t frame_bytes = (size_t)bps * ch;
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return NOMAD_ERR_IO;
    constexpr int64_t kBlockFrames = 1 << 16;  // converted block by block: the scratch stays in L2
    int rc = NOMAD_OK;
    for (int64_t f0 = 0; f0 < wi.frames && rc == NOMAD_OK; f0 += kBlockFrames) {
        const int64_t nf = wi.frames - f0 < kBlockFrames ? wi.frames - f0 : kBlockFrames;
        const size_t want = (size_t)nf * frame_bytes;
        if (raw.size() < want) raw.resize(want);
        size_t got = 0;
        while (got < want) {
            ssize_t r = pread(fd, raw.data() + got, want - got, wi.data_offset + f0 * (int64_t)frame_bytes + (int64_t)got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got != want) {  // the file shrank since the probe
            rc = NOMAD_ERR_IO;
            break;
        }
        const unsigned char* p = raw.data();
        float* o = dst + f0;
        if (ch == 1 && wi.format_tag == 1 && wi.bits == 16) {
            for (int64_t i = 0; i < nf; ++i) o[i] = (float)(int16_t)le16(p + 2 * i) / 32768.0f;
        } else if (ch == 1) {
            for (int64_t i = 0; i < nf; ++i) o[i] = sample_at(p + (size_t)i * bps, wi.format_tag, wi.bits);
        } else {  // the reference averages the first two channels only (nomad.py:199-200)
            for (int64_t i = 0; i < nf; ++i) {
                const unsigned char* q = p + (size_t)i * frame_bytes;
                o[i] = (sample_at(q, wi.format_tag, wi.bits) + sample_at(q + bps, wi.format_tag, wi.bits)) / 2.0f;
            }
        }
    }
    close(fd);
    return rc;
}

// torchaudio's sinc_interp_hann resampling kernel (functional._get_sinc_resample_kernel, lowpass_filter_width 6,
// rolloff 0.99) for orig -> target Hz: `phases` rows of `taps` fp32 coefficients, built in double.
struct ResampleKernel {
    int orig = 1, phases = 1, width = 0, taps = 0;  // reduced rates orig : phases; taps = 2 width + orig
    std::vector<float> k;                           // [phases][taps]
};

inline ResampleKernel make_resample_kernel(int orig_hz, int new_hz) {
    ResampleKernel r;
    const int g = std::gcd(orig_hz, new_hz);
    r.orig = orig_hz / g;
    r.phases = new_hz / g;
    const double lowpass = 6.0, rolloff = 0.99, pi = 3.14159265358979323846;
    const double base = (double)(r.orig < r.phases ? r.orig : r.phases) * rolloff;
    r.width = (int)std::ceil(lowpass * r.orig / base);
    r.taps = 2 * r.width + r.orig;
    r.k.resize((size_t)r.phases * r.taps);
    const double scale = base / r.orig;
    for (int i = 0; i < r.phases; ++i)
        for (int j = 0; j < r.taps; ++j) {
            double t = ((double)(-i) / r.phases + (double)(j - r.width) / r.orig) * base;
            t = t < -lowpass ? -lowpass : (t > lowpass ? lowpass : t);
            const double c = std::cos(t * pi / lowpass / 2.0);
            const double window = c * c;
            t *= pi;
            const double sinc = t == 0.0 ? 1.0 : std::sin(t) / t;
            r.k[(size_t)i * r.taps + j] = (float)(sinc * window * scale);
        }
    return r;
}

inline long long resampled_frames(long long n, int orig_hz, int new_hz) {
    if (orig_hz == new_hz) return n;
    const int g = std::gcd(orig_hz, new_hz);
    const long long o = orig_hz / g, w = new_hz / g;
    return (w * n + o - 1) / o;  // ceil(new * n / orig), as torchaudio
}

// x[0 .. n) at the kernel's source rate -> dst[0 .. resampled_frames): frame f, phase i = sum_j xp[f orig + j] k[i][j] with
// xp = x zero-padded by `width` in front (and behind).
inline void resample_into(const float* x, long long n, const ResampleKernel& rk, float* dst, long long n_out) {
    for (long long o = 0; o < n_out; ++o) {
        const long long f = o / rk.phases;
        const int i = (int)(o - f * rk.phases);
        const float* kk = rk.k.data() + (size_t)i * rk.taps;
        const long long x0 = f * rk.orig - rk.width;  // source index of tap 0
        const int j0 = x0 < 0 ? (int)(-x0) : 0;
        const long long j1 = n - x0 < rk.taps ? n - x0 : rk.taps;
        double acc = 0.0;
        for (long long j = j0; j < j1; ++j) acc += (double)x[x0 + j] * (double)kk[j];
        dst[o] = (float)acc;
    }
}

template <typename Fn>
inline void parallel_for(int n, int threads, Fn fn) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    if (threads <= 1) {
        for (int i = 0; i < n; ++i) fn(i, 0);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> pool;
    pool.reserve((size_t)threads);
    std::atomic<bool> failed{false};  // an exception must not leave a worker thread (std::terminate): remember it, re-throw here
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            try {
                for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i, t);
            } catch (...) {
                failed.store(true);
                next.store(n);
            }
        });
    for (auto& th : pool) th.join();
    if (failed.load()) throw std::runtime_error("a reader thread ran out of memory or failed");
}

}  // namespace wav
}  // namespace nomad
