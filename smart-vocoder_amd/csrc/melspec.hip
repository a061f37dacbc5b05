// Mel front-end of the inference notebook (reference mel_processing.py:51-112):
//   reflect-pad -> STFT(n_fft, hop, hann window, center=False) -> sqrt(re^2 + im^2 + 1e-6)
//   -> Slaney mel filterbank -> log(clamp(., 1e-5))
// Both GEMM-shaped steps run on the MFMA convolution kernel:
//   * n_fft = q*hop, so a frame is q consecutive hop-sized blocks: with the padded waveform laid out as
//     x[c = offset in block][t = block] the windowed DFT is a q-tap convolution with hop input channels and
//     2*(n_fft/2+1) output rows (re_k, im_k), packed as tile pairs so the magnitude is the epilogue (EPI_MAG);
//   * the mel projection is a 1x1 convolution (n_fft/2+1 -> n_mels) with a log-clamp epilogue.
// The filterbank is a restatement of librosa 0.8.0 `filters.mel(sr, n_fft, n_mels, fmin, fmax)` (htk=False,
// norm='slaney'), the third-party dependency the reference calls at mel_processing.py:77; librosa itself is not
// available in the build image, so this constant is checked against its published algorithm only.
#include "svoc_internal.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace svoc {

static ConvArgs mel_args() {
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.pre_slope = 1.0f;
  a.split_row = 1 << 30;
  a.mode = EPI_PLAIN;
  a.out[0].nrows = 1 << 30;
  a.out[1].nrows = 1 << 30;
  a.out[0].div = 1.0f;
  return a;
}

static double hz_to_mel_slaney(double f) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz_slaney(double m) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

// weights[n_mels][1 + n_fft/2], float32 like librosa's default dtype
void slaney_mel_filterbank(int sr, int n_fft, int n_mels, double fmin, double fmax, std::vector<float>& out) {
  const int nb = 1 + n_fft / 2;
  if (fmax <= 0) fmax = sr / 2.0;
  std::vector<double> fftf(nb), melf(n_mels + 2);
  for (int i = 0; i < nb; ++i) fftf[i] = (sr / 2.0) * i / (nb - 1);
  const double m0 = hz_to_mel_slaney(fmin), m1 = hz_to_mel_slaney(fmax);
  for (int i = 0; i < n_mels + 2; ++i) melf[i] = mel_to_hz_slaney(m0 + (m1 - m0) * i / (n_mels + 1));
  out.assign((size_t)n_mels * nb, 0.f);
  for (int i = 0; i < n_mels; ++i) {
    const double fd0 = melf[i + 1] - melf[i], fd1 = melf[i + 2] - melf[i + 1];
    const double enorm = 2.0 / (melf[i + 2] - melf[i]);
    for (int j = 0; j < nb; ++j) {
      const double lower = (fftf[j] - melf[i]) / fd0, upper = (melf[i + 2] - fftf[j]) / fd1;
      const double w = std::max(0.0, std::min(lower, upper));
      out[(size_t)i * nb + j] = (float)(w * enorm);
    }
  }
}

struct MelSpec {
  int n_fft = 0, hop = 0, win = 0, n_mels = 0, nbins = 0, q = 0, pad = 0;
  PackedConv dft, melproj;
  DevBuf ws;

  int create(int n_fft_, int hop_, int win_, int n_mels_, int sr, double fmin, double fmax, hipStream_t st) {
    if (n_fft_ <= 0 || hop_ <= 0 || n_fft_ % hop_ || win_ != n_fft_ || (n_fft_ - hop_) % 2 || n_mels_ <= 0)
      SVOC_FAIL(SVOC_ERR_UNSUPPORTED, "melspec: need n_fft a multiple of hop, win_length == n_fft (got n_fft %d hop %d win %d)", n_fft_, hop_, win_);
    n_fft = n_fft_; hop = hop_; win = win_; n_mels = n_mels_; nbins = n_fft / 2 + 1; q = n_fft / hop; pad = (n_fft - hop) / 2;
    // windowed DFT as conv weights W[row][c][j]: rows [0,nbins) = re, [nbins,2*nbins) = im; sample n = j*hop + c
    std::vector<float> w((size_t)2 * nbins * hop * q);
    const double PI = 3.14159265358979323846;
    for (int k = 0; k < nbins; ++k)
      for (int c = 0; c < hop; ++c)
        for (int j = 0; j < q; ++j) {
          const int n = j * hop + c;
          const double hann = 0.5 - 0.5 * std::cos(2.0 * PI * n / win);          // torch.hann_window (periodic)
          const long long kn = ((long long)k * n) % n_fft;
          const double ang = 2.0 * PI * (double)kn / n_fft;
          w[((size_t)k * hop + c) * q + j] = (float)(hann * std::cos(ang));
          w[((size_t)(nbins + k) * hop + c) * q + j] = (float)(-hann * std::sin(ang));
        }
    std::vector<float> fb;
    slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax, fb);
    DevBuf dw, dfb;
    SVOC_TRY(dw.ensure(w.size() * sizeof(float)));
    SVOC_TRY(dfb.ensure(fb.size() * sizeof(float)));
    SVOC_HIP(hipMemcpy(dw.p, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    SVOC_HIP(hipMemcpy(dfb.p, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice));
    PackSpec sp{}; sp.Cin = hop; sp.Cout = 2 * nbins; sp.K = q; sp.dil = 1; sp.pad = 0; sp.paired = true;
    SVOC_TRY(pack_conv(dft, sp, dw.f(), nullptr, nullptr, st));
    PackSpec mp{}; mp.Cin = nbins; mp.Cout = n_mels; mp.K = 1;
    SVOC_TRY(pack_conv(melproj, mp, dfb.f(), nullptr, nullptr, st));
    return SVOC_OK;
  }

  int frames(long long Lw, int center = 0) const {
    const long long Lp = Lw + 2LL * pad + (center ? 2LL * (n_fft / 2) : 0);
    return Lp < n_fft ? 0 : (int)((Lp - n_fft) / hop + 1);
  }

  // y [B][Lw] in [-1,1] -> spec [B][nbins][F]; center: torch.stft(center=True)'s second reflect padding of n_fft / 2
  int spectrogram(hipStream_t st, const float* y, int B, int Lw, float* spec, int center = 0) {
    if (Lw <= pad) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "melspec: waveform shorter than the reflect padding");
    const int pad2 = center ? n_fft / 2 : 0;
    if (center && Lw + 2LL * pad <= pad2) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "melspec: padded waveform shorter than the centre padding");
    const int F = frames(Lw, center);
    if (F <= 0) return SVOC_OK;
    const int nblk = F + q - 1, ld = round_up(nblk, 4);
    const long long per = (long long)hop * ld;
    SVOC_TRY(ws.ensure((size_t)per * B * sizeof(float)));
    SVOC_TRY(k_frame_blocks(st, y, B, Lw, pad, pad2, hop, ws.f(), per, ld, nblk));
    ConvArgs a = mel_args();
    a.x = ws.f(); a.x_bs = per; a.x_ld = ld; a.Lin = nblk;
    a.Ncols = F;
    a.mode = EPI_MAG;
    a.mag_eps = 1e-6f;
    a.out[0].y = spec; a.out[0].y_bs = (long long)nbins * F; a.out[0].y_ld = F; a.out[0].nrows = nbins;
    return launch_conv(dft, a, B, st);
  }
  // spec [B][nbins][F] -> mel [B][n_mels][F] = log(clamp(basis @ spec, 1e-5))
  int mel(hipStream_t st, const float* spec, int B, int F, float* out) {
    ConvArgs a = mel_args();
    a.x = spec; a.x_bs = (long long)nbins * F; a.x_ld = F; a.Lin = F;
    a.Ncols = F;
    a.log_clamp = 1e-5f;
    a.out[0].y = out; a.out[0].y_bs = (long long)n_mels * F; a.out[0].y_ld = F; a.out[0].nrows = n_mels; a.out[0].flags = F_LOGCLAMP;
    return launch_conv(melproj, a, B, st);
  }
};

}  // namespace svoc

using namespace svoc;
struct svoc_melspec : svoc::HandleDevice { MelSpec m; };

extern "C" {

int svoc_melspec_create(svoc_melspec** out, int n_fft, int hop_length, int win_length, int n_mels, int sampling_rate,
                        double fmin, double fmax) {
  if (!out) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_melspec_create: null argument");
  *out = nullptr;
  try {
    std::unique_ptr<svoc_melspec> h(new svoc_melspec());
    SVOC_TRY(h->m.create(n_fft, hop_length, win_length, n_mels, sampling_rate, fmin, fmax, nullptr));
    *out = h.release();
    return SVOC_OK;
  } catch (const std::exception& e) { ::svoc::set_error("exception: %s", e.what()); return SVOC_ERR_NOMEM; }
}
int svoc_melspec_frames(svoc_melspec* h, int64_t n_samples) { return h ? h->m.frames(n_samples) : 0; }
int svoc_melspec_spectrogram(svoc_melspec* h, void* stream, const float* y, int B, int n_samples, float* spec) {
  if (!h || !y || !spec || B <= 0 || n_samples <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_melspec_spectrogram: bad arguments");
  try { return h->m.spectrogram(as_stream(stream), y, B, n_samples, spec); }
  catch (const std::exception& e) { ::svoc::set_error("exception: %s", e.what()); return SVOC_ERR_NOMEM; }
}
int svoc_melspec_frames_center(svoc_melspec* h, int64_t n_samples, int center) { return h ? h->m.frames(n_samples, center) : 0; }
int svoc_melspec_spectrogram_center(svoc_melspec* h, void* stream, const float* y, int B, int n_samples, int center, float* spec) {
  if (!h || !y || !spec || B <= 0 || n_samples <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_melspec_spectrogram_center: bad arguments");
  try { return h->m.spectrogram(as_stream(stream), y, B, n_samples, spec, center ? 1 : 0); }
  catch (const std::exception& e) { ::svoc::set_error("exception: %s", e.what()); return SVOC_ERR_NOMEM; }
}
int svoc_melspec_mel(svoc_melspec* h, void* stream, const float* spec, int B, int n_frames, float* mel) {
  if (!h || !spec || !mel || B <= 0 || n_frames <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_melspec_mel: bad arguments");
  try { return h->m.mel(as_stream(stream), spec, B, n_frames, mel); }
  catch (const std::exception& e) { ::svoc::set_error("exception: %s", e.what()); return SVOC_ERR_NOMEM; }
}
/* the filterbank constant, for inspection/tests: out [n_mels][1 + n_fft/2] host memory */
int svoc_mel_filterbank(int sampling_rate, int n_fft, int n_mels, double fmin, double fmax, float* out) {
  if (!out || n_fft <= 0 || n_mels <= 0) SVOC_FAIL(SVOC_ERR_INVALID_ARG, "svoc_mel_filterbank: bad arguments");
  std::vector<float> fb;
  slaney_mel_filterbank(sampling_rate, n_fft, n_mels, fmin, fmax, fb);
  memcpy(out, fb.data(), fb.size() * sizeof(float));
  return SVOC_OK;
}
void svoc_melspec_destroy(svoc_melspec* h) { svoc::destroy_handle(h); }

}  // extern "C"
