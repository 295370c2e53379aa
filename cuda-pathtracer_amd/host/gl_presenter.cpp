// gl_presenter.cpp — libptamd_gl.so: the OpenGL presenter of include/ptamd_gl.h (HIP-GL interop through a pixel-unpack
// buffer).  Built only where GL headers exist (`make gl`); never loaded by libptamd.so, the tests' GPU path or bench.py.
#define GL_GLEXT_PROTOTYPES 1
#include <GL/gl.h>
#include <GL/glext.h>

#include "ptamd_gl.h"

#include <hip/hip_runtime.h>
#include <hip/hip_gl_interop.h>

#include <cstdlib>
#include <new>
#include <string>


struct ptamd_gl_presenter {
  uint32_t width = 0, height = 0;
  GLuint pbo = 0, tex = 0, fbo = 0;
  hipGraphicsResource_t res = nullptr;
};

namespace {

thread_local std::string g_gl_error;   // this library's own message slot (ptamd_gl_get_last_error)

int fail(const std::string& msg)
{
  g_gl_error = msg;
  return PTAMD_ERR_ARG;
}

int hip_fail(const char* what, hipError_t e)
{
  g_gl_error = std::string(what) + ": " + hipGetErrorString(e);
  return PTAMD_ERR_HIP;
}

void release(ptamd_gl_presenter* p)
{
  if (p->res) { (void)hipGraphicsUnregisterResource(p->res); p->res = nullptr; }
  if (p->fbo) { glDeleteFramebuffers(1, &p->fbo); p->fbo = 0; }
  if (p->tex) { glDeleteTextures(1, &p->tex); p->tex = 0; }
  if (p->pbo) { glDeleteBuffers(1, &p->pbo); p->pbo = 0; }
}

// interop.cpp:104-116: storage at the new size, registered with the compute API (write-discard: every frame overwrites it)
int allocate(ptamd_gl_presenter* p, uint32_t w, uint32_t h)
{
  release(p);
  p->width = w; p->height = h;
  if (w == 0 || h == 0) return PTAMD_OK;
  while (glGetError() != GL_NO_ERROR) {}
  glGenBuffers(1, &p->pbo);
  glBindBuffer(GL_PIXEL_UNPACK_BUFFER, p->pbo);
  glBufferData(GL_PIXEL_UNPACK_BUFFER, (GLsizeiptr)w * h * 4, nullptr, GL_STREAM_DRAW);
  glBindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
  glGenTextures(1, &p->tex);
  glBindTexture(GL_TEXTURE_2D, p->tex);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
  glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
  glTexImage2D(GL_TEXTURE_2D, 0, GL_RGBA8, (GLsizei)w, (GLsizei)h, 0, GL_RGBA, GL_UNSIGNED_BYTE, nullptr);
  glBindTexture(GL_TEXTURE_2D, 0);
  glGenFramebuffers(1, &p->fbo);
  glBindFramebuffer(GL_READ_FRAMEBUFFER, p->fbo);
  glFramebufferTexture2D(GL_READ_FRAMEBUFFER, GL_COLOR_ATTACHMENT0, GL_TEXTURE_2D, p->tex, 0);
  const GLenum status = glCheckFramebufferStatus(GL_READ_FRAMEBUFFER);
  glBindFramebuffer(GL_READ_FRAMEBUFFER, 0);
  if (glGetError() != GL_NO_ERROR || status != GL_FRAMEBUFFER_COMPLETE) { release(p); return fail("ptamd_gl_presenter: GL object creation failed"); }
  hipError_t e = hipGraphicsGLRegisterBuffer(&p->res, p->pbo, hipGraphicsRegisterFlagsWriteDiscard);
  if (e != hipSuccess) { release(p); return hip_fail("hipGraphicsGLRegisterBuffer", e); }
  return PTAMD_OK;
}

} // namespace

extern "C" {

const char* ptamd_gl_get_last_error(void) { return g_gl_error.c_str(); }

int ptamd_gl_presenter_create(uint32_t width, uint32_t height, ptamd_gl_presenter** out)
{
  if (!out) return fail("ptamd_gl_presenter_create: null out");
  *out = nullptr;
  // without a current context every GL entry point is a no-op that returns 0 / NULL
  const GLubyte* version = glGetString(GL_VERSION);
  if (!version) return fail("ptamd_gl_presenter_create: no current OpenGL context on this thread (the host application creates the "
                            "window and makes its context current, as main.cpp:120-166 does with GLFW); there is no fallback");
  if (std::atoi(reinterpret_cast<const char*>(version)) < 3) return fail("ptamd_gl_presenter_create: OpenGL 3.0 or newer is required");
  ptamd_gl_presenter* p = new (std::nothrow) ptamd_gl_presenter();
  if (!p) return fail("ptamd_gl_presenter_create: out of memory");
  const int rc = allocate(p, width, height);
  if (rc != PTAMD_OK) { delete p; return rc; }
  *out = p;
  return PTAMD_OK;
}

int ptamd_gl_presenter_resize(ptamd_gl_presenter* p, uint32_t width, uint32_t height)
{
  if (!p) return fail("ptamd_gl_presenter_resize: null presenter");
  return allocate(p, width, height);
}

int ptamd_gl_presenter_present(ptamd_gl_presenter* p, const void* surface_rgba8, void* stream)
{
  if (!p || !surface_rgba8) return fail("ptamd_gl_presenter_present: null argument");
  if (p->width == 0 || p->height == 0) return PTAMD_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  // interop.cpp:36-56: map, get the device view, (the reference renders here), unmap
  hipError_t e = hipGraphicsMapResources(1, &p->res, st);
  if (e != hipSuccess) return hip_fail("hipGraphicsMapResources", e);
  void* dst = nullptr;
  size_t bytes = 0;
  e = hipGraphicsResourceGetMappedPointer(&dst, &bytes, p->res);
  if (e == hipSuccess && bytes < (size_t)p->width * p->height * 4) e = hipErrorInvalidValue;
  if (e == hipSuccess) e = hipMemcpyAsync(dst, surface_rgba8, (size_t)p->width * p->height * 4, hipMemcpyDeviceToDevice, st);
  const hipError_t u = hipGraphicsUnmapResources(1, &p->res, st);   // orders GL behind the copy
  if (e != hipSuccess) return hip_fail("ptamd_gl_presenter_present: copy into the interop buffer", e);
  if (u != hipSuccess) return hip_fail("hipGraphicsUnmapResources", u);
  // PBO -> texture on the GPU, then the reference's flipped blit (interop.cpp:67-72): surface row 0 is the top of the picture
  glBindBuffer(GL_PIXEL_UNPACK_BUFFER, p->pbo);
  glBindTexture(GL_TEXTURE_2D, p->tex);
  glPixelStorei(GL_UNPACK_ALIGNMENT, 4);
  glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, (GLsizei)p->width, (GLsizei)p->height, GL_RGBA, GL_UNSIGNED_BYTE, nullptr);
  glBindTexture(GL_TEXTURE_2D, 0);
  glBindBuffer(GL_PIXEL_UNPACK_BUFFER, 0);
  glBindFramebuffer(GL_READ_FRAMEBUFFER, p->fbo);
  glBindFramebuffer(GL_DRAW_FRAMEBUFFER, 0);
  glBlitFramebuffer(0, 0, (GLint)p->width, (GLint)p->height, 0, (GLint)p->height, (GLint)p->width, 0, GL_COLOR_BUFFER_BIT, GL_NEAREST);
  glBindFramebuffer(GL_READ_FRAMEBUFFER, 0);
  return glGetError() == GL_NO_ERROR ? PTAMD_OK : fail("ptamd_gl_presenter_present: GL error during upload / blit");
}

void ptamd_gl_presenter_destroy(ptamd_gl_presenter* p)
{
  if (!p) return;
  release(p);
  delete p;
}

} // extern "C"
