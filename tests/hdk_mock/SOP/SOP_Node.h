// mock: see ../hdk_mock.h (tests only; not the HDK)
#pragma once
#include "../hdk_mock.h"
