#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__global__ void k(const double* in, double* out) {
  const double tx = in[0], ty = in[1], tz = in[2];
  const double dd = tx * tx + ty * ty + tz * tz;
  const double s = sqrt(dd);
  const double scale = 1.0 / s;
  const double dx = tx * scale, dy = ty * scale, dz = tz * scale;
  const double px = in[3], py = in[4], pz = in[5];
  const double a = dx * dx + dz * dz - dy * dy;
  const double b = 2.0 * (px * dx + pz * dz - py * dy);
  const double c = px * px + pz * pz - py * py;
  const double sq = b * b - 4.0 * a * c;
  const double root = sqrt(sq);
  out[0] = dd; out[1] = s; out[2] = scale; out[3] = dx; out[4] = dy; out[5] = dz; out[6] = a; out[7] = b; out[8] = c; out[9] = sq; out[10] = root;
  out[11] = (-b - root) / (2.0 * a); out[12] = (-b + root) / (2.0 * a);
}
static double hx(const char* s) { return strtod(s, nullptr); }
int main() {
  double h[6] = {hx("-0x1.e2b7dddfefa66p-2"), hx("-0x1.e2b7dddfefa64p-2"), 0.0, hx("0x1.aaaaa9e033310p+0"), hx("0x1.d55554f019987p+1"), hx("0x1.77a86166c1deap+0")};
  double *di, *dout, o[13];
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  const char* n[13] = {"dd", "sqrt", "scale", "dx", "dy", "dz", "a", "b", "c", "sq", "root", "r1", "r2"};
  // the same on the host
  double tx = h[0], ty = h[1], tz = h[2];
  volatile double dd = tx * tx + ty * ty + tz * tz; volatile double s = std::sqrt(dd); volatile double scale = 1.0 / s;
  volatile double dx = tx * scale, dy = ty * scale, dz = tz * scale;
  volatile double a = dx * dx + dz * dz - dy * dy, b = 2.0 * (h[3] * dx + h[5] * dz - h[4] * dy), c = h[3] * h[3] + h[5] * h[5] - h[4] * h[4];
  volatile double sq = b * b - 4.0 * a * c, root = std::sqrt(sq);
  double c_[13] = {dd, s, scale, dx, dy, dz, a, b, c, sq, root, (-b - root) / (2.0 * a), (-b + root) / (2.0 * a)};
  for (int i = 0; i < 13; i++) std::printf("%-6s gpu %a   cpu %a   %s\n", n[i], o[i], c_[i], o[i] == c_[i] || (o[i] != o[i] && c_[i] != c_[i]) ? "" : "DIFFERENT");
  return 0;
}
