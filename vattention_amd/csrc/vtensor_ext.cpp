// Tiny torch binding: wrap a reserved (possibly unmapped) device virtual range as a strided
// torch.Tensor WITHOUT touching it.  Plays the role of the reference's alloc_vtensor /
// VirtualTensorAllocator (/root/reference/vattention/vtensor.h:9-125): no-op deleter (the range is
// freed by cleanup()), device type "cuda" (which names the HIP backend on torch-ROCm), and — new —
// explicit strides so the page-rounded per-request stride is visible.
// Passing target_device keeps at::from_blob from querying pointer attributes of an unmapped VA.
#include <torch/extension.h>

#include <vector>

static void noop_deleter(void*) {}

static at::Tensor tensor_from_va(uint64_t ptr, std::vector<int64_t> sizes, std::vector<int64_t> strides,
                                 py::object dtype, int device) {
    TORCH_CHECK(sizes.size() == strides.size(), "sizes and strides must have the same rank");
    const at::ScalarType st = torch::python::detail::py_object_to_dtype(dtype);
    const caffe2::TypeMeta meta = c10::scalarTypeToTypeMeta(st);
    const c10::Device dev(c10::kCUDA, static_cast<c10::DeviceIndex>(device));
    // bytes spanned by the strided view
    int64_t span = 1;
    for (size_t i = 0; i < sizes.size(); i++) {
        TORCH_CHECK(sizes[i] > 0 && strides[i] > 0, "sizes and strides must be positive");
        span += (sizes[i] - 1) * strides[i];
    }
    const size_t nbytes = static_cast<size_t>(span) * meta.itemsize();
    void* p = reinterpret_cast<void*>(ptr);
    // Built by hand (no at::from_blob): nothing here may call into the GPU runtime or look the
    // pointer up — the range is reserved but not yet backed by memory.
    c10::DataPtr dp(p, p, &noop_deleter, dev);
    auto storage = c10::make_intrusive<c10::StorageImpl>(c10::StorageImpl::use_byte_size_t(), nbytes, std::move(dp),
                                                         /*allocator=*/nullptr, /*resizable=*/false);
    auto t = at::detail::make_tensor_base<c10::TensorImpl>(std::move(storage), c10::DispatchKeySet(c10::DispatchKey::CUDA), meta);
    t.unsafeGetTensorImpl()->set_sizes_and_strides(sizes, strides);
    return t;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.def("tensor_from_va", &tensor_from_va, "strided tensor over a raw device virtual address (never dereferenced)");
}
