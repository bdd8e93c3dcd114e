// scannerCompat.h -- the few third-party names the reference's public headers leak
// (fft.h:3 <fftw3.h>, process.h:5 <gnuradio/fft/window.h>), so that code written against
// the reference's class surface compiles against this host library unchanged.
#pragma once
#include <cstdint>

typedef float fftwf_complex[2];  // FFTW's single-precision complex: {re, im}

namespace gr {
namespace fft {
namespace window {
// gr::fft::window::win_type (GNU Radio 3.7/3.8 numbering).  scan.cpp:215 only ever passes
// WIN_BLACKMAN_HARRIS; every type is accepted (process.cpp:18 hands any of them to window::build).
enum win_type {
  WIN_HAMMING = 0,
  WIN_HANN = 1,
  WIN_BLACKMAN = 2,
  WIN_RECTANGULAR = 3,
  WIN_KAISER = 4,
  WIN_BLACKMAN_HARRIS = 5,
  WIN_BLACKMAN_hARRIS = 5,
  WIN_BARTLETT = 6,
  WIN_FLATTOP = 7
};
}  // namespace window
}  // namespace fft
}  // namespace gr
