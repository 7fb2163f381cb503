// Force-included (-include) in front of the REFERENCE's unmodified cpu/ROIAlign_cpu.cpp when oracle/build_ref.sh compiles it
// against the torch headers of this image (2.10): the source calls AT_DISPATCH_FLOATING_TYPES(input.type(), ...) and the
// macro's ::detail::scalar_type() no longer has an overload for at::DeprecatedTypeProperties.  Test
// infrastructure only; no reference code is copied.
#pragma once
#include <torch/extension.h>
namespace detail {
inline at::ScalarType scalar_type(const at::DeprecatedTypeProperties& t) { return t.scalarType(); }
}  // namespace detail
