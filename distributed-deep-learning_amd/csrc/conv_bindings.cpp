// Bindings of the convolution / batch-norm engine kernels (filled in as kernels land).
#include <torch/extension.h>

void register_conv_ops(pybind11::module_& m) { (void)m; }
