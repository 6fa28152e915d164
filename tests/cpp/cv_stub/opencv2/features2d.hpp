// TEST-ONLY stand-in (see core.hpp in this directory): cv::KeyPoint lives in core.hpp here.
#include "core.hpp"
