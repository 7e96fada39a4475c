// fused-operator kernels of degree 8: instantiates the variant dispatch of bp5_device.hpp for this degree
#include "bp5_device.hpp"
template int apply_degree_impl<8>(bp5_mf *, const double *, const double *, double *, uint32_t, uint32_t, bool);
