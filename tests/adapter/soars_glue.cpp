// soars_glue.cpp -- the ONE translation unit INTEGRATION.md asks a SOARS maintainer to add, compiled here against the mock
// headers of tests/adapter/soars_stub/ so that the RTS_ADAPTER_WITH_SOARS branch of rts_adapter.hpp (rts_amd::SoarsTraits and
// the definition of rs::RTS with the reference's signature, ray_tracer.cpp:512) is parsed, type-checked and linked.
#include "rsworld.cuh"
#include "rsradar.cuh"
#include "rstarget.cuh"
#include "rsparameters.cuh"
#include "rsresponse.cuh"
#include "rspath.cuh"
#define RTS_ADAPTER_WITH_SOARS
#include "rts_adapter.hpp"

int main(int argc, char**)
{
    void (*entry)(rs::World*, unsigned int, unsigned int) = &rs::RTS;     // the reference's signature
    if (argc > 100) { rs::World w; entry(&w, 1024, 65535); }               // never executed by the test: it only has to link
    return entry ? 0 : 1;
}
