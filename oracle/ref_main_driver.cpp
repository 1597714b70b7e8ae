// ref_main_driver.cpp - TEST INFRASTRUCTURE, built only where /root/reference exists (oracle/Makefile: ref-main).
//
// Links the reference's OWN orchestration - Controleur/PathTracer.cpp and Controleur/PathTracer_Importer.cpp, compiled
// unmodified where they lie - against the product's backend shim (csrc/PathTracer_HIP.cpp built against the reference's
// own headers, -DPTMI_USE_REFERENCE_HEADERS) and libptmi.so, and calls PathTracer_Main with the ten arguments of
// Maya/RayTracer.cpp:121.  What this file adds is only what the reference keeps in files that need Windows or an SDK:
//   * an importer (the reference's are Maya / LumenRT / a Windows-only file reader): reads the scene dump the tests write;
//   * the "window" class the orchestration paints into (Alone/PathTracer_Dialog.cpp needs Win32 / Maya): keeps a copy of
//     the last image it was handed, which is how the result leaves PathTracer_Main (it frees everything it owns);
//   * the scene-cache exporter's symbols (Controleur/PathTracer_FileImporter.cpp uses fopen_s): present, never called.
//
// usage: ref_main_driver scene.bin out.bin [numImages]      exit code 0 = PathTracer_Main returned true
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "PathTracer.h"
#include "PathTracer_FileImporter.h"

namespace PathTracerNS
{
	static const char* g_out_path = nullptr;
	static unsigned g_paint_calls = 0;

	// ---- window ------------------------------------------------------------------------------------------------------
	const char* PathTracerDialog::exportFolderPath = "";
	PathTracerDialog::PathTracerDialog() : pathTracerWidth(0), pathTracerHeight(0), imageIndex(0), saveRenderedImages(false) {}
	bool PathTracerDialog::PaintWindow(RGBAColor const* imageColor, float const* imageRay)
	{
		g_paint_calls++;
		imageIndex++;
		if (!g_out_path) return true;
		FILE* o = std::fopen(g_out_path, "wb");
		if (!o) return false;
		const unsigned meta[3] = {g_paint_calls, pathTracerWidth, pathTracerHeight};
		const size_t n = (size_t)pathTracerWidth * pathTracerHeight;
		std::fwrite(meta, 4, 3, o);
		std::fwrite(imageColor, sizeof(RGBAColor), n, o);
		std::fwrite(imageRay, sizeof(float), n, o);
		std::fclose(o);
		return true;
	}
	void PathTracerDialog::PaintTexture(Uchar4 const*, Texture&) {}

	// ---- scene-cache exporter: symbols only ---------------------------------------------------------------------------
	const std::string PathTracerFileImporter::fileSizesPath = "";
	const std::string PathTracerFileImporter::filePointersPath = "";
	const std::string PathTracerFileImporter::fileTextureDataPath = "";
	PathTracerFileImporter::PathTracerFileImporter() {}
	void PathTracerFileImporter::Import(uint, uint, bool) { throw std::runtime_error("scene cache files are not part of this harness"); }
	void PathTracerFileImporter::Export() { throw std::runtime_error("scene cache files are not part of this harness"); }

	// ---- importer: the scene dump of tests/ (same layout as tests/shim_driver.cpp reads) --------------------------------
	// u32 W,H,depth,sampler,nImages, nTri,nLights,nMat,nTex,nTexels; Float4 camPos,camDir,camRight,camUp; Sky; Triangle[];
	// Light[]; Material[]; Texture[]; Uchar4[]
	class DumpImporter : public PathTracerImporter
	{
	public:
		explicit DumpImporter(const char* path) : path_(path) {}
		template <class T> static T* read_array(FILE* f, size_t n)
		{
			T* p = new T[n ? n : 1];  // PathTracer_Clear delete[]s every array
			if (n && std::fread(p, sizeof(T), n, f) != n) throw std::runtime_error("scene dump is truncated");
			return p;
		}
		virtual void Import(uint image_width, uint image_height, bool)
		{
			FILE* f = std::fopen(path_, "rb");
			if (!f) throw std::runtime_error("cannot open the scene dump");
			unsigned h[10];
			if (std::fread(h, 4, 10, f) != 10) throw std::runtime_error("scene dump is truncated");
			*ptr__global__imageWidth = image_width;
			*ptr__global__imageHeight = image_height;
			*ptr__global__imageSize = image_width * image_height;
			*ptr__global__triangulationSize = h[5];
			*ptr__global__lightsSize = h[6];
			*ptr__global__materiauxSize = h[7];
			*ptr__global__texturesSize = h[8];
			*ptr__global__texturesDataSize = h[9];
			if (std::fread(ptr__global__cameraPosition, sizeof(Float4), 1, f) != 1 || std::fread(ptr__global__cameraDirection, sizeof(Float4), 1, f) != 1 ||
				std::fread(ptr__global__cameraRight, sizeof(Float4), 1, f) != 1 || std::fread(ptr__global__cameraUp, sizeof(Float4), 1, f) != 1 ||
				std::fread(ptr__global__sky, sizeof(Sky), 1, f) != 1)
				throw std::runtime_error("scene dump is truncated");
			*ptr__global__triangulation = read_array<Triangle>(f, h[5]);
			*ptr__global__lights = read_array<Light>(f, h[6]);
			*ptr__global__materiaux = read_array<Material>(f, h[7]);
			*ptr__global__textures = read_array<Texture>(f, h[8]);
			*ptr__global__texturesData = read_array<Uchar4>(f, h[9]);
			std::fclose(f);
		}
	private:
		const char* path_;
	};
}

int main(int argc, char** argv)
{
	using namespace PathTracerNS;
	if (argc < 3) return 2;
	FILE* f = std::fopen(argv[1], "rb");
	if (!f) return 2;
	unsigned h[10];
	if (std::fread(h, 4, 10, f) != 10) return 2;
	std::fclose(f);
	g_out_path = argv[2];
	const uint numImages = argc > 3 ? (uint)std::atoi(argv[3]) : h[4];
	PathTracer_SetImporter(new DumpImporter(argv[1]));
	// Maya/RayTracer.cpp:121: PathTracer_Main(width, height, numImages, saveRenderedImages, loadSky, exportScene, sampler, rayMaxDepth, printLog, superSampling)
	const bool ok = PathTracer_Main(h[0], h[1], numImages, false, false, false, (Sampler)h[3], h[2], false, false);
	std::printf("PathTracer_Main returned %s after %u window updates\n", ok ? "true" : "false", g_paint_calls);
	return ok ? 0 : 1;
}
