// CPU timeline of the host layer as a chrome://tracing / Perfetto JSON file, switched on by the environment variable the
// reference uses: GRANITE_TIMELINE_TRACE=<path> (threading/thread_group.cpp:174; util/timeline_trace_file.hpp:35-94 is the
// reference's writer and its GRANITE_SCOPED_TIMELINE_EVENT macro).  Same event shape -- "B" / "E" pairs with name, tid and pid as
// strings and ts in microseconds -- so a trace of the HIP executor loads beside one of a Vulkan build.  Built differently: no writer
// thread and no event pool; a scope costs two clock reads and an append to a buffer of its own thread, and the buffers are written
// out by flush() (gra_sync, application teardown, process exit).  With the variable unset a scope is one predictable branch.
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

namespace Granite
{
class TimelineTrace
{
public:
	static TimelineTrace &get()
	{
		static TimelineTrace trace;
		return trace;
	}
	bool enabled() const { return file != nullptr; }

	struct Event
	{
		const char *name; // string literal or a string that outlives the trace (pass names are interned below)
		uint64_t begin_ns, end_ns;
	};

	// Names of graph passes and the like: kept alive until the process ends.
	const char *intern(const std::string &name)
	{
		std::lock_guard<std::mutex> holder{lock};
		for (auto &s : interned)
			if (*s == name)
				return s->c_str();
		interned.push_back(std::make_unique<std::string>(name));
		return interned.back()->c_str();
	}

	static uint64_t now_ns()
	{
		return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
	}

	void set_thread_name(const char *name) { local().tid = name; }

	void record(const char *name, uint64_t begin_ns, uint64_t end_ns)
	{
		auto &buffer = local();
		std::lock_guard<std::mutex> holder{buffer.lock}; // uncontended except against flush()
		buffer.events.push_back({name, begin_ns, end_ns});
	}

	// Appends what the threads have recorded so far to the file.
	void flush()
	{
		if (!file)
			return;
		std::lock_guard<std::mutex> holder{lock};
		for (auto &buffer : buffers)
		{
			std::vector<Event> events;
			{
				std::lock_guard<std::mutex> inner{buffer->lock};
				events.swap(buffer->events);
			}
			// A scope is recorded when it EXITS, i.e. children before their parents.  The file carries "B" / "E" pairs in time order
			// per thread, properly nested, as the reference's writer emits them while the scopes run (util/timeline_trace_file.cpp):
			// scopes sorted by begin (the longer one first on a tie = the parent), ends emitted as soon as the next begin lies past them.
			std::stable_sort(events.begin(), events.end(), [](const Event &a, const Event &b) {
				return a.begin_ns != b.begin_ns ? a.begin_ns < b.begin_ns : a.end_ns > b.end_ns;
			});
			auto emit = [&](const Event &e, char phase, uint64_t ns) {
				fprintf(file, "{ \"name\": \"%s\", \"ph\": \"%c\", \"tid\": \"%s\", \"pid\": \"0\", \"ts\": %f },\n", e.name, phase, buffer->tid.c_str(),
				        double(int64_t(ns - base_ns)) * 1e-3);
			};
			std::vector<const Event *> open;
			for (auto &e : events)
			{
				while (!open.empty() && open.back()->end_ns <= e.begin_ns && !(open.back()->end_ns == e.begin_ns && open.back()->begin_ns == e.begin_ns))
				{
					emit(*open.back(), 'E', open.back()->end_ns);
					open.pop_back();
				}
				emit(e, 'B', e.begin_ns);
				open.push_back(&e);
			}
			while (!open.empty())
			{
				emit(*open.back(), 'E', open.back()->end_ns);
				open.pop_back();
			}
		}
		fflush(file);
	}

	~TimelineTrace()
	{
		flush();
		if (file)
		{
			// the trailing comma is fine for the viewers (the reference leaves its array open as well); close it for strict parsers
			fputs("{ \"name\": \"end-of-trace\", \"ph\": \"i\", \"tid\": \"main\", \"pid\": \"0\", \"ts\": 0, \"s\": \"g\" }\n]\n", file);
			fclose(file);
		}
	}

private:
	struct ThreadBuffer
	{
		std::mutex lock;
		std::string tid;
		std::vector<Event> events;
	};
	TimelineTrace()
	{
		if (const char *path = getenv("GRANITE_TIMELINE_TRACE"))
		{
			file = fopen(path, "w");
			if (!file)
				fprintf(stderr, "[granite-hip] GRANITE_TIMELINE_TRACE: cannot open %s\n", path);
			else
				fputs("[\n", file);
		}
		base_ns = now_ns();
	}
	ThreadBuffer &local()
	{
		static thread_local ThreadBuffer *mine = nullptr;
		if (!mine)
		{
			std::lock_guard<std::mutex> holder{lock};
			buffers.push_back(std::make_unique<ThreadBuffer>());
			mine = buffers.back().get();
			mine->tid = buffers.size() == 1 ? "main" : "thread-" + std::to_string(buffers.size() - 1);
		}
		return *mine;
	}
	FILE *file = nullptr;
	uint64_t base_ns = 0;
	std::mutex lock;
	std::vector<std::unique_ptr<ThreadBuffer>> buffers;
	std::vector<std::unique_ptr<std::string>> interned;
};

class ScopedTimelineEvent
{
public:
	explicit ScopedTimelineEvent(const char *name_) : name(name_)
	{
		if (TimelineTrace::get().enabled())
			begin_ns = TimelineTrace::now_ns();
	}
	~ScopedTimelineEvent()
	{
		if (begin_ns)
			TimelineTrace::get().record(name, begin_ns, TimelineTrace::now_ns());
	}
	ScopedTimelineEvent(const ScopedTimelineEvent &) = delete;
	void operator=(const ScopedTimelineEvent &) = delete;

private:
	const char *name;
	uint64_t begin_ns = 0;
};
} // namespace Granite

#define GRANITE_TIMELINE_CONCAT_(a, b) a##b
#define GRANITE_TIMELINE_CONCAT(a, b) GRANITE_TIMELINE_CONCAT_(a, b)
// util/timeline_trace_file.hpp:88-94
#define GRANITE_SCOPED_TIMELINE_EVENT(name) ::Granite::ScopedTimelineEvent GRANITE_TIMELINE_CONCAT(timeline_scope_, __LINE__){name}
