// Include-path shim: the reference's test/test_dse.cpp:1 still includes <Spark/...>.
#include "../cask/Dse.hpp"
