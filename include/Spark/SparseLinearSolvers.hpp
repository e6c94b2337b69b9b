// Include-path shim: the reference's test/test_bicg.cpp:1 still includes <Spark/...>.
#include "../cask/SparseLinearSolvers.hpp"
