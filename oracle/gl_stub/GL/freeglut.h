// see glew.h in this directory (test infrastructure stub)
#pragma once
#include "glew.h"
