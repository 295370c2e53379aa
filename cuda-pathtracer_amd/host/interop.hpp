// interop.hpp — C++ host mirror of the reference's presentation surface, without OpenGL.
//
// The reference renders into one of two GL renderbuffers registered with CUDA (driver::Interop,
// cuda_opengl/include/driver/interop.h:18-80, src/driver/interop.cpp): GPUProcessor::render() does
//     _interop.map(stream); raytrace(_interop.getArray(), ...); _interop.unmap(stream);
// and main.cpp:198-200 then calls  interop.blit(); interop.swap();  — blit copies the front buffer to the
// window flipped vertically (interop.cpp:67-72), swap alternates the two buffers.
//
// Same member names and meaning here; what backs them differs because a MI355X node has no display:
//   * the two buffers are linear RGBA8 device surfaces (ptamd_device_alloc), row 0 = top of the picture, which is
//     what ptamd_raytrace writes — so the vertical flip of the GL blit is already done and blit() is a straight copy;
//   * map()/unmap() have nothing to register and return PTAMD_OK (they keep the call sequence of render() intact);
//   * blit() brings the current buffer to the host (after the stream's pending work) and hands it to the presenter
//     callback given at construction — a headless host writes a file (ptamd_image_save_png) or does nothing; a host
//     with a window builds with -DPTAMD_WITH_GL and calls blit(gl_presenter, stream) instead: the optional
//     libptamd_gl.so (include/ptamd_gl.h) uploads the surface through a HIP-registered pixel buffer and does the
//     reference's flipped glBlitFramebuffer;
//   * errors are ptamd_status codes instead of cudaError_t.
// Header-only; link with -lptamd.
#pragma once

#include "ptamd.h"
#ifdef PTAMD_WITH_GL
#include "ptamd_gl.h"   // optional OpenGL presenter (libptamd_gl.so) for hosts that own a window
#endif

#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace ptamd_host {

class Interop
{
public:
  // presenter(pixels, width, height): RGBA8, row 0 = top, alpha = 0 as the kernel writes it (raytrace.cu:232)
  using Presenter = std::function<void(const uint8_t*, unsigned int, unsigned int)>;

  Interop(ptamd_context* ctx, unsigned int w, unsigned int h, Presenter presenter = Presenter())
    : _ctx(ctx), _width(0), _half_width(0), _height(0), _half_height(0), _index(0), _presenter(presenter)
  {
    _d_surface[0] = _d_surface[1] = nullptr;
    setSize(w, h);
  }
  Interop(const Interop&) = delete;
  Interop& operator=(const Interop&) = delete;
  ~Interop() { clean(); }

  int map(void* /*stream*/) { return PTAMD_OK; }
  int unmap(void* /*stream*/) { return PTAMD_OK; }

  int clean()
  {
    for (int i = 0; i < 2; ++i) {
      if (_d_surface[i]) ptamd_device_free(_ctx, _d_surface[i]);
      _d_surface[i] = nullptr;
    }
    return PTAMD_OK;
  }

  /// Swaps the framebuffers for double buffering.
  void swap() { _index = (_index + 1) % 2; }

  /// The reference clears the current framebuffer to white (interop.cpp:58-63).
  int clear(void* stream = nullptr)
  {
    return _d_surface[_index] ? ptamd_device_memset(_ctx, _d_surface[_index], 0xFF, bytes(), stream) : PTAMD_OK;
  }

  /// Copies the current framebuffer to the "screen": here, to the presenter callback.
  int blit(void* stream = nullptr)
  {
    if (!_d_surface[_index] || bytes() == 0) return PTAMD_OK;
    _host.resize(bytes());
    int rc = ptamd_device_to_host(_ctx, _host.data(), _d_surface[_index], bytes(), stream);
    if (rc == PTAMD_OK && _presenter) _presenter(_host.data(), _width, _height);
    return rc;
  }

#ifdef PTAMD_WITH_GL
  /// With a window (-DPTAMD_WITH_GL, -lptamd_gl): the reference's blit (interop.cpp:67-72) — the current framebuffer goes
  /// to the default GL framebuffer through the HIP-GL interop buffer of the presenter, without a host copy.
  int blit(ptamd_gl_presenter* gl, void* stream = nullptr)
  {
    if (!_d_surface[_index] || bytes() == 0) return PTAMD_OK;
    return ptamd_gl_presenter_present(gl, _d_surface[_index], stream);
  }
#endif

  int setSize(const unsigned int w, const unsigned int h)
  {
    clean();
    _width = w; _half_width = (unsigned int)(w * 0.5); _height = h; _half_height = (unsigned int)(h * 0.5);
    if (bytes() == 0) return PTAMD_OK;
    for (int i = 0; i < 2; ++i) {
      int rc = ptamd_device_alloc(_ctx, bytes(), &_d_surface[i]);
      if (rc == PTAMD_OK) rc = ptamd_device_memset(_ctx, _d_surface[i], 0, bytes(), nullptr);
      if (rc != PTAMD_OK) { clean(); return rc; }
    }
    return PTAMD_OK;
  }

  inline int getIndex() { return _index; }
  /// The surface raytrace() writes (replaces the cudaArray of the mapped renderbuffer).
  inline void* getArray() { return _d_surface[_index]; }
  void getSize(unsigned int& w, unsigned int& h) { w = _width; h = _height; }
  inline unsigned width() const { return _width; }
  inline unsigned half_width() const { return _half_width; }
  inline unsigned height() const { return _height; }
  inline unsigned half_height() const { return _half_height; }
  /// Host copy made by the last blit().
  const std::vector<uint8_t>& pixels() const { return _host; }

private:
  size_t bytes() const { return (size_t)_width * _height * 4; }

  ptamd_context* _ctx;
  unsigned int _width, _half_width, _height, _half_height;
  int _index;
  void* _d_surface[2];
  std::vector<uint8_t> _host;
  Presenter _presenter;
};

} // namespace ptamd_host
