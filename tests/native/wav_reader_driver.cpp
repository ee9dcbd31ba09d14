// Host-only driver for nomad_amd/csrc/wav_reader.h, built by tests/test_wav_native.py with -fsanitize=address,undefined:
// probes every file given on the command line, decodes / resamples it to 16 kHz into an exactly-sized heap buffer (so that
// any out-of-bounds access trips the sanitizer) and prints "<status> <rate> <channels> <frames> <frames at 16 kHz> <sum>".
#include <cstdio>
#include <vector>

#include "../../nomad_amd/csrc/wav_reader.h"

int main(int argc, char** argv) {
    using namespace nomad::wav;
    for (int a = 1; a < argc; ++a) {
        nomad_wav_info wi;
        const int st = probe_one(argv[a], &wi);
        if (st != NOMAD_OK) {
            printf("%d 0 0 0 0 0\n", st);
            continue;
        }
        std::vector<unsigned char> raw;
        std::vector<float> mono((size_t)wi.frames);
        int rc = read_one(argv[a], wi, mono.data(), raw);
        const long long n16 = resampled_frames(wi.frames, wi.sample_rate, 16000);
        std::vector<float> out((size_t)n16);
        if (rc == NOMAD_OK && wi.sample_rate != 16000) {
            const ResampleKernel rk = make_resample_kernel(wi.sample_rate, 16000);
            resample_into(mono.data(), wi.frames, rk, out.data(), n16);
        } else if (rc == NOMAD_OK) {
            out = mono;
        }
        double sum = 0;
        for (float v : out) sum += v;
        printf("%d %d %d %lld %lld %.9g\n", rc, wi.sample_rate, wi.channels, wi.frames, n16, sum);
    }
    return 0;
}
