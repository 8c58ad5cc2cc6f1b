// Pool scene fixture entry point (filled in with the Pool variant; see lrl/Scenes.hpp).
#include <string>
namespace { struct Handle; }
extern "C" int lrlh_scene_pool(void*, const char*, int) { return 1; }
