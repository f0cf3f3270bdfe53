/*
 * lorastencil_ref_shims.h -- the reference's seven operator prototypes (C++ linkage), implemented by
 * liblorastencil_shims.so on top of the C ABI in lorastencil.h.
 *
 * These are exactly the declarations of /root/reference/src/{1d/1d_utils.h:45-47, 2d/2d_utils.h:47-51,
 * 3d/3d_utils.h:44-48}: same names, argument order and meaning, `void` return, and the reference's error
 * policy (print + exit(1), 2d_utils.h:22-36).  A reference harness (src/{1d,2d,3d}/main.cu) links against
 * the shim library unchanged.
 */
#ifndef LORASTENCIL_REF_SHIMS_H
#define LORASTENCIL_REF_SHIMS_H

void gpu_1d1r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
              const int time, const int input_n);
void gpu_1d2r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
              const int time, const int input_n);

void gpu_star_2d1r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                   const int times, const int input_m, const int input_n);
void gpu_star_2d3r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                   const int times, const int input_m, const int input_n);
void gpu_box_2d3r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                  const int times, const int input_m, const int input_n);

void gpu_box_3d1r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                  const int times, const int input_h, const int input_m, const int input_n);
void gpu_star_3d1r(const double *__restrict__ in, double *__restrict__ out, const double *__restrict__ params,
                   const int times, const int input_h, const int input_m, const int input_n);

#endif
