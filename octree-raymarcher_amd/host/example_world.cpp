// example_world.cpp — the reference's main-loop calls (src/Main.cpp:80-81,222,314-319,340-367) against svo::World.
//   g++ -std=c++17 -I. example_world.cpp -L.. -lsvo_amd -Wl,-rpath,'$ORIGIN/..' -o example_world
#include <cstdio>
#include "svo_world.hpp"

int main(int argc, char **argv)
{
    const int depth = argc > 1 ? std::atoi(argv[1]) : 8;
    try {
        svo::World world;
        world.init(4, 1, 4, 128, depth);                 // Main.cpp:80   world.init(4, 4, 4, 128) (one y layer holds the terrain)
        world.load_gpu(0);                               // Main.cpp:81
        svo::Camera cam({ 256.0f, 150.0f, -40.0f }, { 0.0f, -0.5f, 0.866f }, { 0.0f, 1.0f, 0.0f }, 60.0f, 640, 360);
        svo::GBuffer gbuffer;
        world.draw(cam, gbuffer, /*shadow=*/true);       // Main.cpp:190-222  shadow pass + primary pass
        svo_stream_synchronize(nullptr);
        size_t hits = 0, shadowed = 0;
        for (const svo_hit &h : gbuffer.download()) { hits += h.flags & SVO_HIT_FLAG; shadowed += (h.flags & SVO_SHADOWED) != 0; }
        // two views (the frame's camera and one a step to the right) in one launch: svo_trace_frames
        svo::Camera cam2({ 262.0f, 150.0f, -40.0f }, { 0.0f, -0.5f, 0.866f }, { 0.0f, 1.0f, 0.0f }, 60.0f, 640, 360);
        svo::GBuffer pair;
        world.draw_frames({ cam, cam2 }, pair, /*shadow=*/true);
        svo_stream_synchronize(nullptr);
        const std::vector<svo_hit> both = pair.download(), first = gbuffer.download();
        if (std::memcmp(both.data(), first.data(), first.size() * sizeof(svo_hit)) != 0) { std::fprintf(stderr, "draw_frames: first view differs from draw\n"); return 3; }
        // Main.cpp:317 computeTarget.  (From exactly x = 256 this direction lies in a voxel-lattice plane: the
        // reference's 0 * inf = NaN makes that ray a miss, SURVEY.md App. C — so the cursor starts a little off it.)
        svo::vec3 sigma = { 0.0f, 0.0f, 0.0f };
        const bool hit = svo::chunkmarch({ 250.3f, 150.0f, -40.0f }, { 0.0f, -0.5f, 0.866f }, &world, &sigma);
        std::printf("hits %zu shadowed %zu cursor %s (%.3f %.3f %.3f)\n", hits, shadowed, hit ? "hit" : "miss", sigma.x, sigma.y, sigma.z);
        // Main.cpp:350-357 build(): a cube of material 5 at the cursor, in every chunk it overlaps; then the edit cursor again
        if (hit) {
            const svo::vec3 cmin = { sigma.x - 4.0f, sigma.y, sigma.z - 4.0f }, cmax = { sigma.x + 4.0f, sigma.y + 8.0f, sigma.z + 4.0f };
            for (int i = 0; i < world.volume; ++i) world.build(i, cmin, cmax, 5);
            svo::vec3 again = { 0.0f, 0.0f, 0.0f };
            const bool hit2 = svo::chunkmarch({ 250.3f, 150.0f, -40.0f }, { 0.0f, -0.5f, 0.866f }, &world, &again);
            if (!hit2 || !(again.y > sigma.y)) { std::fprintf(stderr, "build: the cursor ray does not stop on the new cube\n"); return 4; }
            for (int i = 0; i < world.volume; ++i) world.destroy(i, cmin, cmax);       // Main.cpp:340-347 destroy()
            world.shift({ 1, 0, 0 });                    // World::shift, src/World.cpp:334-378
            world.draw(cam, gbuffer, /*shadow=*/true);
            svo_stream_synchronize(nullptr);
            std::printf("after build / destroy / shift: cursor moved up by %.3f\n", again.y - sigma.y);
        }
        // the same view as the fragment shader marches it (shaders/Chunkmarch.glsl: EPS 1/4096, BIGEPS guard, LEAF hits at t)
        world.semantics = SVO_SEMANTICS_GLSL;
        svo::GBuffer glsl;
        world.draw(cam, glsl, /*shadow=*/true);
        svo_stream_synchronize(nullptr);
        size_t ghits = 0;
        for (const svo_hit &h : glsl.download()) ghits += (h.flags & SVO_HIT_FLAG) != 0;
        std::printf("GLSL semantics: %zu hits\n", ghits);
        if (ghits == 0) return 5;
        return hits > 0 ? 0 : 1;
    } catch (const svo::Error &e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 2;
    }
}
