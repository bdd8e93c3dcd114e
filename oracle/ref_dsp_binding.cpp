// ref_dsp_binding.cpp -- C binding around the reference's utility.cpp, compiled from where it lies.
//
// TEST INFRASTRUCTURE ONLY (see scn_oracle.h).  utility.cpp holds the in-tree arithmetic of the hot path: the three
// integer -> complex-float converters (utility.cpp:9-84, rows a1-a3 of SURVEY.md section 8) and the log-magnitude map
// (utility.cpp:86-98, row a6).  It includes "fft.h" -> <fftw3.h> for one thing, the type fftwf_complex.  This image has no
// FFTW, but ROCm ships an implementation of the FFTW3 API -- hipFFTW, /opt/rocm/include/hipfft/hipfftw.h, "include it
// instead of fftw3.h" -- so the recipe (oracle/Makefile, target ref) puts that header on the include path UNDER THE NAME
// the reference asks for (a symlink made at build time inside the git-ignored oracle/_ref/; no declaration is written by
// this repo) and passes `-include cmath`: utility.cpp calls log2 / sqrt without including a header for them and relies on
// one arriving transitively, which GCC 11's <algorithm> no longer provides.  With <cmath> in scope the
// log is the double one -- the case the oracle's default mode restates.
// Nothing of the reference is copied; this file only calls what utility.h declares.
#include <cstdint>

#include <fftw3.h>  // = hipfftw.h, see above

#include "utility.h"  // -I/root/reference

extern "C" {

void ref_short_planar_to_float(int16_t *re, int16_t *im, float *dst, uint32_t n, uint32_t enob, int correct_dc) {
  Utility::short_complex_to_float_complex(re, im, reinterpret_cast<fftwf_complex *>(dst), n, enob, correct_dc != 0);
}

void ref_short_complex_to_float(int16_t *iq, float *dst, uint32_t n, uint32_t enob, int correct_dc) {
  Utility::short_complex_to_float_complex(reinterpret_cast<int16_t(*)[2]>(iq), reinterpret_cast<fftwf_complex *>(dst), n, enob,
                                          correct_dc != 0);
}

void ref_byte_complex_to_float(int8_t *iq, float *dst, uint32_t n, uint32_t enob, int correct_dc) {
  Utility::byte_complex_to_float_complex(reinterpret_cast<int8_t(*)[2]>(iq), reinterpret_cast<fftwf_complex *>(dst), n, enob,
                                         correct_dc != 0);
}

void ref_complex_to_magnitude(float *spectrum, float *magnitudes, uint32_t n) {
  Utility::complex_to_magnitude(reinterpret_cast<fftwf_complex *>(spectrum), magnitudes, n);
}

}  // extern "C"
