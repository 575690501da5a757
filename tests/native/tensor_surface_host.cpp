// TEST INFRASTRUCTURE ONLY: the Tensor-API scenarios of tests/tensor_surface.inc and the input-pipeline
// probes of tests/data_reader.inc compiled against
// this repo's host layer (visiontransformer_amd/host/include, the reference's header paths) and
// linked with libvit_host.so, the way an application of the reference would be.
#include "core/Tensor.hpp"
#define TENSOR_SURFACE_FN vith_test_tensor_scenario
#include "../tensor_surface.inc"

#include "utils/DataReader.hpp"
#define DATA_READER_FN vith_test_data_reader
#define CLASS_WEIGHTS_FN vith_test_class_weights
#include "../data_reader.inc"
