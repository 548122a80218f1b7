"""pixell_amd: MI355X-native backend for pixell's harmonic-transform hot path.

  pixell_amd.curvedsky  map2alm / alm2map / alm_info ... (mirror of pixell/curvedsky.py's SHT API)
  pixell_amd.sht        ducc0.sht.experimental-shaped functions (synthesis_2d, analysis_2d, ...)
  pixell_amd.fft        fft / ifft / rfft / irfft + the `hip` engine object for pixell.fft.engines
  pixell_amd.enmap      the few enmap pieces the path needs (ndmap, fullsky_geometry, fft, ifft)
All arithmetic runs in hand-written HIP kernels (pixell_amd/csrc) behind include/pxsht.h.
"""
__version__ = "0.1.0"
