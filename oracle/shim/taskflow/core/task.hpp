#pragma once
#include "../taskflow.hpp"
